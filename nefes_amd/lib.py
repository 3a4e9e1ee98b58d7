"""ctypes binding of libnefes_hip.so (C ABI: include/nefes_hip.h).

There is no CPU fallback: if the library cannot be loaded the product path raises.
`import torch` happens first so that the HIP runtime already mapped by PyTorch-ROCm
(same SONAME libamdhip64.so.7) is the one the kernels launch on, which makes
`torch.cuda.current_stream().cuda_stream` a valid hipStream_t for every call.
"""
import ctypes as C
import os

import torch  # noqa: F401  (must precede dlopen of libnefes_hip.so, see module docstring)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NEFES_HIP_LIB") or os.path.join(_HERE, "libnefes_hip.so")   # override = debugging builds only


class NefesNetDesc(C.Structure):
    _fields_ = [("width", C.c_int32), ("feat_dim", C.c_int32), ("has_transient", C.c_int32), ("xyz_encoding", C.c_int32)]


class NefesStreamInfo(C.Structure):
    _fields_ = [("slab_off", C.c_uint64), ("n_slabs", C.c_uint32), ("bias_floats", C.c_uint32), ("bias_off", C.c_uint64),
                ("scale_off", C.c_uint32), ("scale_count", C.c_uint32)]


class NefesBlobInfo(C.Structure):
    _fields_ = [("total_bytes", C.c_uint64), ("stream", NefesStreamInfo * 13)]


class NefesHashGridDesc(C.Structure):
    _fields_ = [("n_levels", C.c_int32), ("n_features", C.c_int32), ("log2_hashmap_size", C.c_int32),
                ("base_resolution", C.c_int32), ("per_level_scale", C.c_float), ("bound", C.c_float)]


ABI_VERSION = 14       # NEFES_ABI_VERSION of include/nefes_hip.h
STREAM_FWD_SIGMA, STREAM_FWD_STATIC, STREAM_FWD_FULL, STREAM_BWD_FULL, STREAM_FWD_SIGMA_X6, STREAM_FWD_FULL_X6, STREAM_BWD_FULL_X6, STREAM_BWD_STATIC = 0, 1, 2, 3, 4, 5, 6, 7
STREAM_FWD_SIGMA_H3, STREAM_FWD_FULL_H3, STREAM_BWD_FULL_H3, STREAM_FWD_STATIC_H3, STREAM_BWD_STATIC_H3 = 8, 9, 10, 11, 12
FIELD_SIGMA, FIELD_STATIC, FIELD_FULL = 0, 1, 2
XYZ_FREQ10, XYZ_EXTERNAL32 = 0, 1
(TB_E, TB_DV, TB_L1, TB_FINAL, TB_DIR, TB_T0, TB_T1, TB_T2, TB_RGB, TB_SIG, TB_TH, TB_END) = (0, 1, 2, 10, 11, 12, 13, 14, 15, 16,
                                                                                              17, 18)
COMP_TRANSIENT, COMP_STATIC_ONLY, COMP_SIGMA_ONLY, COMP_WHITE_BKGD, COMP_FEAT_WEIGHTS_ONLY = 1, 2, 4, 8, 16

_p, _i, _f, _u32, _sz = C.c_void_p, C.c_int, C.c_float, C.c_uint32, C.c_size_t
_desc = C.POINTER(NefesNetDesc)

# name -> (restype, argtypes); mirrors include/nefes_hip.h declaration by declaration
SIGNATURES = {
    "nefes_version": (_i, []),
    "nefes_blob_info": (_i, [_desc, C.POINTER(NefesBlobInfo)]),
    "nefes_stream_slab_bytes": (_sz, [_desc, _i]),
    "nefes_fusion_input_fwd": (_i, [_i, _i, _i, _p, _p, _p, C.POINTER(_f), C.POINTER(_f), _p, _p, _p]),
    "nefes_fusion_input_bwd": (_i, [_i, _i, _i, _p, _p, _p, C.POINTER(_f), _p, _p, _p]),
    "nefes_upcos_loss_fwd": (_i, [_i, _i, _i, _i, _i, _i, _p, _p, _p, _p, _p]),
    "nefes_upcos_loss_bwd": (_i, [_i, _i, _i, _i, _i, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _p, _p, _p]),
    "nefes_bicubic_gather_table": (_i, [_i, _i, _i, _i, _i, _p, _p, _p, _p]),
    "nefes_bicubic_gram": (_i, [_i, _i, _i, _i, _p, _p]),
    "nefes_bn_train_fwd": (_i, [_i, _i, C.c_int64, _i, _p, _p, _p, C.c_double, C.c_double, _p, _p, _p, _p, _p, _p]),
    "nefes_bn_train_bwd": (_i, [_i, _i, C.c_int64, _i, _p, _p, _p, _p, _p, _p]),
    "nefes_svd_reg_fwd": (_i, [_i, _p, _p, _p, _p]),
    "nefes_svd_reg_bwd": (_i, [_i, _p, _p, _p, _p]),
    "nefes_regressed_pose_fwd": (_i, [_i, _p, _i, _f, _f, _f, _f, _p, _p, _p]),
    "nefes_regressed_pose_bwd": (_i, [_i, _p, _i, _f, _p, _p, _p]),
    "nefes_upcos_prepare": (_i, [_i, _i, _i, _i, _i, _i, _p, _p, _p, _p, _p, _p, _p, _i, _p, _p, _p, _p]),
    "nefes_upcos_gram_fwd": (_i, [_i, _i, _i, _p, _p, _p, _p, _p, _i, _p, _p, _p, _p]),
    "nefes_upcos_gram_bwd": (_i, [_i, _i, _i, _p, _p, _p, _p, _p, _p]),
    "nefes_upcos_gram_lds_bytes": (_sz, [_i, _i, _i]),
    "nefes_adam_step": (_i, [_i, _p, _p, _p, _p, _p, _p, C.c_double, C.c_double, C.c_double, _p]),
    "nefes_pack_weights": (_i, [_desc, C.POINTER(_p), _i, _p, _sz]),
    "nefes_pack_map": (_i, [_desc, _p, _sz, _p]),
    "nefes_pack_h3_plan": (_i, [_desc, _p, _sz, C.POINTER(_sz)]),
    "nefes_pack_device": (_i, [_p, C.c_int64, _p, C.c_int64, _p, _i, _p, _p, _p]),
    "nefes_raygen_fwd": (_i, [_i, _i, _f, _p, _i, _i, _p, _p, _p, _p]),
    "nefes_raygen_bwd_workspace": (_sz, [_i]),
    "nefes_raygen_bwd": (_i, [_i, _i, _f, _p, _i, _i, _p, _p, _p, _p, _p, _p]),
    "nefes_ndc_fwd": (_i, [_i, _i, _f, _f, _i, _p, _p, _p, _p, _p]),
    "nefes_ndc_bwd": (_i, [_i, _i, _f, _f, _i, _p, _p, _p, _p, _p, _p, _p]),
    "nefes_coarse_depths": (_i, [_i, _i, _f, _f, _i, _p, _p, _p, _p]),
    "nefes_coarse_depths_rays": (_i, [_i, _i, _p, _i, _i, _p, _p, _p, _p]),
    "nefes_field_mask_bytes": (_sz, [_desc, C.c_int64]),
    "nefes_field_fwd": (_i, [_desc, _p, _i, _i, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p]),
    "nefes_field_bwd": (_i, [_desc, _p, _i, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p]),
    "nefes_ray_grad_reduce": (_i, [_i, _i, _p, _p, _p, _p, _p, _p, _p]),
    "nefes_composite_fwd": (_i, [_i, _i, _i, _u32, _f, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p]),
    "nefes_composite_bwd": (_i, [_i, _i, _i, _u32, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p]),
    "nefes_hashgrid_table_entries": (_sz, [C.POINTER(NefesHashGridDesc)]),
    "nefes_hashgrid_fwd": (_i, [C.POINTER(NefesHashGridDesc), _p, C.c_int64, _p, _p, _p]),
    "nefes_hashgrid_bwd_x": (_i, [C.POINTER(NefesHashGridDesc), _p, C.c_int64, _p, _p, _p, _p]),
    "nefes_field_bwd_static": (_i, [_desc, _p, _i, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p]),
    "nefes_field_bwd_x6": (_i, [_desc, _p, _i, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p]),
    "nefes_field_fwd_x6": (_i, [_desc, _p, _i, _i, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p]),
    "nefes_field_bwd_h3": (_i, [_desc, _p, _i, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p]),
    "nefes_field_bwd_static_h3": (_i, [_desc, _p, _i, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p]),
    "nefes_feat_head_fwd": (_i, [_i, _i, _i, _p, _p, _p, _p, _p]),
    "nefes_feat_head_bwd": (_i, [_i, _i, _i, _p, _p, _p, _p, _p]),
    "nefes_field_fwd_h3_fh": (_i, [_desc, _p, _i, _i, _i, _p, _p, _p, _p, _p, _p, _p]),
    "nefes_field_bwd_h3_fh": (_i, [_desc, _p, _i, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p]),
    "nefes_field_fwd_h3_hashgrid": (_i, [_desc, _p, C.POINTER(NefesHashGridDesc), _p, _i, _i, _i, _p, _p, _p, _i, _p, _p, _p, _p]),
    "nefes_field_bwd_h3_hashgrid": (_i, [_desc, _p, C.POINTER(NefesHashGridDesc), _p, _i, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p]),
    "nefes_field_fwd_h3": (_i, [_desc, _p, _i, _i, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p]),
    "nefes_field_fwd_h3_zrow": (_i, [_desc, _p, _i, _i, _i, _p, _p, _p, _p, _p, _p, _p]),
    "nefes_coarse_sample": (_i, [_i, _i, _i, _p, _p, _i, _p, _i, _p, _p, _p, _p]),
    "nefes_conv2d_same": (_i, [_i, _i, _i, _i, _i, _i, _p, _p, _p, _p, _i, _p, _p]),
    "nefes_probe_mfma_clock": (_i, [_i, _i, C.POINTER(C.c_double), C.POINTER(C.c_double), _p]),
    "nefes_probe_store_hazard": (_i, [_p, C.c_int64, _i, _p]),
    "nefes_probe_pk_mul": (_i, [_p, C.c_int64, _i, _p]),
    "nefes_probe_hazard": (_i, [_i, _i, _i, _i, _p, _p]),
    "nefes_probe_aggressor": (_i, [_i, _i, _i, _p, _p]),
    "nefes_train_rows": (_sz, [_desc]),
    "nefes_train_row_offset": (_i, [_desc, _i]),
    "nefes_field_fwd_train": (_i, [_desc, _p, _i, _i, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p]),
    "nefes_field_fwd_train_h3": (_i, [_desc, _p, _i, _i, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p]),
    "nefes_field_bwd_train": (_i, [_desc, _p, _i, _i, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p]),
    "nefes_field_bwd_train_h3": (_i, [_desc, _p, _i, _i, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p]),
    "nefes_train_head_grad": (_i, [_desc, _i, _i, _i, _p, _p, _p, _p]),
    "nefes_train_dx": (_i, [C.c_int64, _i, _p, _i, _i, _p, _i, _i, _p, _i, _i, _i, _p, _p]),
    "nefes_train_dw": (_i, [C.c_int64, _i, _p, _i, _i, _p, _i, _i, _i, _i, _p, _p]),
    "nefes_train_dw_bias": (_i, [C.c_int64, _i, _p, _i, _i, _p, _i, _i, _i, _i, C.c_int64, _p, _p]),
    "nefes_bicubic_up_fwd": (_i, [C.c_int64, _i, _i, _i, _i, _i, _i, _i, _i, _p, _p, _p]),
    "nefes_bicubic_up_bwd": (_i, [C.c_int64, _i, _i, _i, _i, _i, _i, _i, _i, _p, _p, _p, _p]),
    "nefes_pose_compose_fwd": (_i, [_i, _p, _p, _p, C.c_float, C.POINTER(C.c_float), C.c_float, _p, _p]),
    "nefes_pose_compose_bwd": (_i, [_i, _p, _p, _p, C.c_float, C.POINTER(C.c_float), C.c_float, _p, _p, _p, _p]),
    "nefes_psnr_ssim_workspace": (_sz, [_i, _i, _i]),
    "nefes_psnr_ssim": (_i, [_i, _i, _i, _p, C.c_int64, C.c_int64, _p, C.c_int64, C.c_int64, _p, _p, _p]),
    "nefes_cosine_loss_scratch_doubles": (_sz, [_i]),
    "nefes_cosine_loss_fwd": (_i, [_i, C.c_int64, _p, _p, _p, _p, _p]),
    "nefes_cosine_loss_bwd": (_i, [_i, C.c_int64, _p, _p, _p, _p, _p, _p]),
    "nefes_sample_pdf_merge": (_i, [_i, _i, _i, _i, _p, _p, _p, _i, _p, _p, _p, _p, _p, _p]),
}

_lib = None


def load():
    """Return the loaded library; raise RuntimeError (never fall back) if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C nefes_amd/csrc`. nefes_amd has no CPU or PyTorch fallback for the hot path.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)      # AttributeError here = ABI mismatch; let it propagate
        fn.restype = res
        fn.argtypes = args
    if lib.nefes_version() != ABI_VERSION:
        raise RuntimeError("libnefes_hip.so ABI version mismatch")
    _lib = lib
    return lib


COMPILED_SET = ("fp16 two-part instances (default): widths 128 / 256 x feature heads of 0..29 or 30..141 channels with the frequency "
                "embedding, width 256 x 0..29 channels with an external 32-feature embedding; bf16x6 and fp32-MFMA instances "
                "(NEFES_SPLIT=x6 / f32): width 256 x 16 channels and width 128 x 128 channels only")


def check(rc, what):
    if rc != 0:
        kind = {-1: "bad argument", -2: "unsupported configuration", -3: "bad weight blob"}.get(rc, f"hipError {rc}")
        raise RuntimeError(f"{what} failed: {kind}" + (f".  Compiled: {COMPILED_SET}" if rc == -2 else ""))
