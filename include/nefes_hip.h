/* nefes_hip.h -- C ABI of libnefes_hip.so: the MI355X (gfx950) implementation of the NeFeS
 * volumetric render-and-refine hot path.
 *
 * The reference (ActiveVisionLab/NeFeS) has no native/FFI boundary: its seam is the Python call
 * surface of script/models/{ray_utils,rendering,nerfh_nff}.py.  Each entry point below replaces the
 * torch op sequence of the reference function cited next to it (paths relative to the reference
 * root).  The Python side (nefes_amd/) binds these with ctypes and wraps them in
 * torch.autograd.Function; see INTEGRATION.md for the binding a reference maintainer would add.
 *
 * Conventions
 *  - every pointer marked "dev" is device memory owned by the caller (e.g. tensor.data_ptr());
 *    the library never allocates, frees or retains device memory and keeps no global state;
 *  - all kernels are asynchronous on `stream` (a hipStream_t passed as void*); no host sync inside;
 *  - return value: 0 on success, a positive hipError_t from the launch, or a negative NEFES_E_*
 *    argument error; never aborts, never throws;
 *  - fp32 everywhere; "raw_t" is the per-sample field output stored channel-major per ray,
 *    raw_t[N][R][S]  (the reference's raw[N][S][R] is raw_t.permute(0,2,1));
 *  - N rays, S samples per ray, M = N*S samples, C feature channels, Wd MLP width.
 */
#ifndef NEFES_HIP_H
#define NEFES_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NEFES_ABI_VERSION 14

#define NEFES_E_BADARG (-1)     /* null pointer / non-positive size */
#define NEFES_E_UNSUPPORTED (-2) /* width / feat_dim / sample count outside the compiled set */
#define NEFES_E_BADBLOB (-3)

/* field network description: NeRFH_NFF ctor arguments (script/models/nerfh_nff.py:421-427).
 * depth is fixed at 8 with the skip at layer 5 (skips=[4]), encodings at L=10 / L=4. */
typedef struct NefesNetDesc {
    int32_t width;         /* Wd: 128 or 256 */
    int32_t feat_dim;      /* C: f_dim (16 or 128 compiled; 3+C <= 160) */
    int32_t has_transient; /* 1 for the 'fine' net (encode_transient), 0 for 'coarse' */
    int32_t xyz_encoding;  /* NEFES_XYZ_FREQ10: 63 frequency features computed in-kernel from the sample position;
                            * NEFES_XYZ_EXTERNAL32: 32 features supplied per sample (xyz_enc), e.g. nefes_hashgrid_fwd */
} NefesNetDesc;
#define NEFES_XYZ_FREQ10 0
#define NEFES_XYZ_EXTERNAL32 1

/* where each weight stream lives inside a packed blob (all offsets in bytes from the blob start) */
typedef struct NefesStreamInfo {
    uint64_t slab_off;   /* first slab */
    uint32_t n_slabs;    /* nefes_stream_slab_bytes(desc, stream) bytes each (16 / 32 / 48 KiB: nefes_amd/csrc/layout.h) */
    uint32_t bias_floats;
    uint64_t bias_off;   /* bias block (fp32, natural row order per layer); 0 if none */
    uint32_t scale_off;  /* _H3 streams: word index, inside the bias block, of the weight-scale exponent table (int32 per
                            segment in stream order: the segment's weights are stored multiplied by 2^e); counted in bias_floats */
    uint32_t scale_count;
} NefesStreamInfo;

typedef struct NefesBlobInfo {
    uint64_t total_bytes;
    NefesStreamInfo stream[13]; /* indexed by NEFES_STREAM_* in csrc/layout.h; n_slabs == 0 if absent */
} NefesBlobInfo;

/* compositing variants of raw2outputs_NeRFH_NFF (script/models/nerfh_nff.py:25-166) */
#define NEFES_COMP_TRANSIENT 1u   /* raw carries the 5 transient channels (output_transient=True) */
#define NEFES_COMP_STATIC_ONLY 2u /* variant B: test_time and not transient_at_test (:92-117) */
#define NEFES_COMP_SIGMA_ONLY 4u  /* variant D: coarse net at test time, raw = sigma only (:33-35,83-89) */
#define NEFES_COMP_WHITE_BKGD 8u  /* :126-127 */
#define NEFES_COMP_FEAT_WEIGHTS_ONLY 16u /* nefes_composite_bwd only: instead of the C rows w_s[s] g_feat[c] of the feature channels' gradient, write the
                                          * static weight w_s[s] into the FIRST feature channel's row (the factored head's backward forms the
                                          * products itself: nefes_field_bwd_h3_fh with g_gmap) */

/* field forward modes (run_network_NeRFH_NFF branches, script/models/nerfh_nff.py:192-231) */
#define NEFES_FIELD_SIGMA 0  /* coarse + test_time: sigma only, R = 1 */
#define NEFES_FIELD_STATIC 1 /* output_transient=False: R = 3+C+1 */
#define NEFES_FIELD_FULL 2   /* output_transient=True:  R = 3+C+6 */

int nefes_version(void);

/* ---- weights ------------------------------------------------------------------------------- */
/* Blob geometry for a network description. */
int nefes_blob_info(const NefesNetDesc* desc, NefesBlobInfo* info);
/* bytes of one slab of weight stream `stream` (NEFES_STREAM_*) for this network description; 0 for an unknown stream */
size_t nefes_stream_slab_bytes(const NefesNetDesc* desc, int stream);
/* Host-side layout transform of a NeRFH_NFF state_dict into the MFMA fragment streams the kernels
 * consume.  `tensors` = host fp32 pointers, (weight, bias) per layer in the reference's construction
 * order (nerfh_nff.py:452-505): xyz_encoding_1..8, xyz_encoding_final, dir_encoding.0,
 * static_sigma.0, static_rgb.0 [, transient_encoding.0/.2/.4, transient_sigma.0, transient_rgb.0,
 * transient_beta.0]; torch layout [out, in] row-major.  `blob` = host buffer of info.total_bytes. */
int nefes_pack_weights(const NefesNetDesc* desc, const float* const* tensors, int n_tensors, void* blob,
                       size_t blob_bytes);
/* Re-packing on the device, for training (script/run_nefes.py:42-108: the weights change at every optimizer.step()).
 * nefes_pack_map (host): for every 16-bit slot of the blob, map[slot] = (flat index + 1) << 3 | part, where the flat index
 * counts the elements of the tensor table above concatenated in order (tensor_elems_out[i] = elements of tensor i, may
 * be NULL), part 0/1 = low/high half of the fp32 value, 2/3/4 = bf16 hi/mid/lo of the bf16x6 streams, code 0 = zero.
 * n_entries >= total_bytes / 2.
 * The fp16 two-part streams: parts 5/6 = fp16 hi/lo of value * 2^e with the matrix's exponent group in bits 27..31 of the
 * code, part 7 = a half of the exponent word; code 0xffffffff = word written by the plan's reductions.
 * nefes_pack_h3_plan (host): the reductions those streams need from the parameter vector (one exponent per weight matrix,
 * row bounds and bias maxima of the scale tables); `needed` = ints required (query with plan = NULL); plan[0] = n_jobs.
 * nefes_pack_device: blob[slot] = part(flat[...]) for every slot after the 512-byte header; flat, map, plan, scratch
 * (>= 32 ints), blob on the device; blob must have been initialised once by nefes_pack_weights.  plan = NULL leaves the fp16
 * streams as they are.  Result bit-identical to nefes_pack_weights. */
int nefes_pack_map(const NefesNetDesc* desc, uint32_t* map, size_t n_entries, int64_t* tensor_elems_out);
int nefes_pack_h3_plan(const NefesNetDesc* desc, int32_t* plan, size_t n_ints, size_t* needed);
int nefes_pack_device(const float* flat, int64_t n_params, const uint32_t* map, int64_t n_entries, const int32_t* plan,
                      int n_jobs, int32_t* scratch, void* blob, void* stream);

/* ---- rays (script/models/ray_utils.py) ------------------------------------------------------- */
/* get_rays (:5-16) + viewdirs = d/|d| (rendering.py:217) for image rows [row0, row0+nrows).
 * c2w: dev, 12 floats row-major 3x4.  Outputs dev [nrows*W, 3]. */
int nefes_raygen_fwd(int H, int W, float focal, const float* c2w, int row0, int nrows, float* rays_o, float* rays_d,
                     float* viewdirs, void* stream);
/* backward of the above: g_c2w[12] (dev) = sum over rays.  g_* may be NULL (treated as zero).
 * workspace: dev, >= nefes_raygen_bwd_workspace(nrows*W) bytes. */
size_t nefes_raygen_bwd_workspace(int n_rays);
int nefes_raygen_bwd(int H, int W, float focal, const float* c2w, int row0, int nrows, const float* g_rays_o,
                     const float* g_rays_d, const float* g_viewdirs, void* workspace, float* g_c2w, void* stream);
/* ndc_rays (:27-44) and its backward to (rays_o, rays_d). */
int nefes_ndc_fwd(int H, int W, float focal, float near, int n, const float* rays_o, const float* rays_d, float* out_o,
                  float* out_d, void* stream);
int nefes_ndc_bwd(int H, int W, float focal, float near, int n, const float* rays_o, const float* rays_d,
                  const float* g_out_o, const float* g_out_d, float* g_rays_o, float* g_rays_d, void* stream);
/* coarse depths: z = near*(1-t)+far*t or the lindisp form, optional stratified jitter with a
 * caller-supplied t_rand[N,Nc] (rendering.py:96-112).  t: dev [Nc] = torch.linspace(0,1,Nc). */
int nefes_coarse_depths(int N, int Nc, float near, float far, int lindisp, const float* t, const float* t_rand,
                        float* z, void* stream);
/* same with per-ray bounds, as render_rays reads them from the packed ray batch (rendering.py:90-93, columns 6:8 of
 * the [n, 8+3+hist] batch built at :227-235): near = bounds[ray*stride], far = bounds[ray*stride + 1] (dev, floats). */
int nefes_coarse_depths_rays(int N, int Nc, const float* bounds, int stride, int lindisp, const float* t,
                             const float* t_rand, float* z, void* stream);

/* ---- field MLP (nerfh_nff.py:168-231 run_network + :234-270 Embedder + :525-576 forward) ------- */
/* Either (rays_o, rays_d, z) are given and pts = o + d*z is formed in-kernel (rendering.py:114,142),
 * or pts[M,3] is given (run_network call surface), or (NEFES_XYZ_EXTERNAL32) xyz_enc[M,32] is given.
 * viewdirs [N,3] is ignored in SIGMA mode.
 * raw_t: dev [N][R][S].  masks: dev uint32 [ceil(M/32)][mask_words][64] or NULL (FULL mode only;
 * needed by nefes_field_bwd).  packed: dev blob from nefes_pack_weights. */
size_t nefes_field_mask_bytes(const NefesNetDesc* desc, int64_t M);
int nefes_field_fwd(const NefesNetDesc* desc, const void* packed, int mode, int N, int S, const float* rays_o,
                    const float* rays_d, const float* z, const float* pts, const float* xyz_enc, const float* viewdirs,
                    float* raw_t, uint32_t* masks, void* stream);
/* backward to the inputs (frozen weights, no dW): g_pts [M,3] (NEFES_XYZ_FREQ10) or g_xyz_enc [M,32]
 * (NEFES_XYZ_EXTERNAL32; positions are then not needed), and g_viewdirs_s [M,3] (per sample). */
int nefes_field_bwd(const NefesNetDesc* desc, const void* packed, int N, int S, const float* rays_o,
                    const float* rays_d, const float* z, const float* pts, const float* viewdirs, const float* raw_t,
                    const float* g_raw_t, const uint32_t* masks, float* g_pts, float* g_xyz_enc, float* g_viewdirs_s,
                    void* stream);
/* per-ray reduction of the above: g_o = sum_s g_pts, g_d = sum_s z*g_pts, g_v = sum_s g_viewdirs_s. */
int nefes_ray_grad_reduce(int N, int S, const float* z, const float* g_pts, const float* g_viewdirs_s, float* g_rays_o,
                          float* g_rays_d, float* g_viewdirs, void* stream);

/* ---- compositing (nerfh_nff.py:25-166) -------------------------------------------------------- */
/* Outputs may be NULL when not wanted.  weights: [N,S] (variant B: the static-only weights). */
int nefes_composite_fwd(int N, int S, int C, uint32_t flags, float beta_min, const float* raw_t, const float* z,
                        float* rgb, float* feat, float* disp, float* acc, float* depth, float* weights, float* beta,
                        void* stream);
/* g_* upstream gradients may be NULL (zero).  g_raw_t: dev [N][R][S], fully written. */
int nefes_composite_bwd(int N, int S, int C, uint32_t flags, const float* raw_t, const float* z, const float* g_rgb,
                        const float* g_feat, const float* g_disp, const float* g_acc, const float* g_depth,
                        const float* g_weights, const float* g_beta, float* g_raw_t, void* stream);

/* ---- hierarchical sampling (rendering.py:23-66 sample_pdf, :132-141 z_mid / sort) -------------- */
/* layout 0 (render_rays): z_coarse [N,Nc] coarse depths and weights [N,Nc] coarse compositing weights;
 *   bins = z_mid and the weights[...,1:-1] slice are formed inside (rendering.py:132-134).
 * layout 1 (sample_pdf call surface): z_coarse = bins [N,Nc-1], weights = [N,Nc-2] exactly as the
 *   reference function receives them; z_fine must be NULL.
 * u: dev [Ni] (u_per_ray = 0, e.g. torch.linspace(0,1,Ni)) or [N,Ni] (u_per_ray = 1).  cdf_in (optional,
 * [N,Nc-1]) overrides the internally built CDF (stage test: indices bit-exact on identical CDFs).
 * Outputs: z_fine [N,Nc+Ni] ascending (or NULL), z_samples [N,Ni], inds int32 [N,Ni], cdf_out [N,Nc-1]. */
int nefes_sample_pdf_merge(int N, int Nc, int Ni, int layout, const float* z_coarse, const float* weights,
                           const float* u, int u_per_ray, const float* cdf_in, float* z_fine, float* z_samples,
                           int32_t* inds, float* cdf_out, void* stream);
/* The coarse pass behind its field kernel as ONE launch: compositing variant D (the sigma-only coarse pass' weights,
 * nerfh_nff.py:83-89) + sample_pdf (rendering.py:23-66) + sort(cat[z_vals, z_samples]) (:132-141), Nc = 64 / 128 / 256, Nc + Ni <= 512.
 * sigma [N,Nc] = raw_t of nefes_field_fwd* in NEFES_FIELD_SIGMA mode; z = [N,Nc] depths, or ONE row [Nc] shared by every ray
 * (z_shared_row = 1); u as in nefes_sample_pdf_merge.  Outputs: z_fine [N,Nc+Ni]; optional z_samples [N,Ni] and weights_out [N,Nc]
 * (= nefes_composite_fwd's `weights`).  Bit-identical to nefes_composite_fwd(NEFES_COMP_SIGMA_ONLY) + nefes_sample_pdf_merge.
 * NEFES_E_UNSUPPORTED for other Nc (use those two calls). */
int nefes_coarse_sample(int N, int Nc, int Ni, const float* sigma, const float* z, int z_shared_row, const float* u, int u_per_ray,
                        float* z_fine, float* z_samples, float* weights_out, void* stream);

/* ---- multiresolution hash-grid encoding (script/models/nerfh_tcnn.py:60-75,151-156; tiny-cuda-nn semantics) ---- */
/* PARITY UNPINNED: the reference delegates this to tiny-cuda-nn (not vendored, not version-pinned) and its own model
 * using it is orphaned; checked only against oracle/hashgrid_ref.py's restatement of the published algorithm. */
typedef struct NefesHashGridDesc {
    int32_t n_levels;          /* 16 */
    int32_t n_features;        /* 2 (only value built) */
    int32_t log2_hashmap_size; /* 19 */
    int32_t base_resolution;   /* 16 */
    float per_level_scale;     /* exp(ln(2048/16)/15) */
    float bound;               /* scene bound: x01 = (x+bound)/(2 bound) */
} NefesHashGridDesc;
size_t nefes_hashgrid_table_entries(const NefesHashGridDesc* desc);   /* entries of n_features floats; 0 = unsupported */
/* x [M,3] -> enc [M, n_levels*n_features]; table: dev fp32 [entries][n_features]. */
int nefes_hashgrid_fwd(const NefesHashGridDesc* desc, const float* table, int64_t M, const float* x, float* enc,
                       void* stream);
/* backward to the positions (frozen table): g_x [M,3]. */
int nefes_hashgrid_bwd_x(const NefesHashGridDesc* desc, const float* table, int64_t M, const float* x, const float* g_enc,
                         float* g_x, void* stream);

/* nefes_field_fwd(mode = NEFES_FIELD_SIGMA or NEFES_FIELD_FULL) with the hidden 256x256 products (layers 2..8 and
 * xyz_encoding_final) as bf16x6 split products on v_mfma_f32_32x32x16_bf16 -- exact hi/mid/lo bf16 triples, six cross terms,
 * fp32 accumulation: fp32-level accuracy (nefes_amd/csrc/field_fwd_x6.hip).  Width 256, C = 16, either xyz encoding;
 * same outputs and the same ReLU-mask words as nefes_field_fwd, so nefes_field_bwd follows unchanged. */
int nefes_field_fwd_x6(const NefesNetDesc* desc, const void* packed, int mode, int N, int S, const float* rays_o,
                       const float* rays_d, const float* z, const float* pts, const float* xyz_enc, const float* viewdirs,
                       float* raw_t, uint32_t* masks, void* stream);

/* nefes_field_bwd for a NEFES_FIELD_STATIC forward (raw channels rgb+feature, sigma; masks from nefes_field_fwd in that mode):
 * what autograd does for the coarse network in train mode (rendering.py:116-125 with test_time=False). */
int nefes_field_bwd_static(const NefesNetDesc* desc, const void* packed, int N, int S, const float* rays_o,
                           const float* rays_d, const float* z, const float* pts, const float* viewdirs, const float* raw_t,
                           const float* g_raw_t, const uint32_t* masks, float* g_pts, float* g_viewdirs_s, void* stream);

/* nefes_field_bwd (width 256, C = 16) with the transposed products as bf16x6 split
 * products; consumes the mask words of either forward kernel. */
int nefes_field_bwd_x6(const NefesNetDesc* desc, const void* packed, int N, int S, const float* rays_o, const float* rays_d,
                       const float* z, const float* pts, const float* viewdirs, const float* raw_t, const float* g_raw_t,
                       const uint32_t* masks, float* g_pts, float* g_xyz_enc, float* g_viewdirs_s, void* stream);

/* The same two functions (nefes_field_fwd in NEFES_FIELD_SIGMA / NEFES_FIELD_FULL -- and, frequency embedding only, NEFES_FIELD_STATIC --
 * mode, nefes_field_bwd) with the products as
 * fp16 TWO-PART split products on v_mfma_f32_32x32x16_f16 -- (hi, lo) fp16 pairs of power-of-two scaled operands, three cross
 * terms hh + hl + lh, fp32 accumulation: 22 significant bits per operand, fp32-level accuracy at half the matrix-core work of
 * the _x6 calls (nefes_amd/csrc/field_h3.h; replaces script/models/nerfh_nff.py:168-231,525-576 + autograd like they do).
 * Weights are scaled per matrix by the packer (exponent table in the NEFES_STREAM_*_H3 streams), activations / gradient vectors
 * per sample and product inside the kernels.  Shapes (round 4): widths 128 / 256 x the two head CLASSES of csrc/layout.h --
 * nefes_head_class: C <= 29 (e.g. BASELINE's 16 channels) and 30 <= C <= 141 (the reference's FEATURE_DIM = 128, nerfh_nff.py:21) --
 * with the frequency embedding, C itself a run-time value (desc->feat_dim); width 256 / C <= 29 with an external embedding.
 * NEFES_E_UNSUPPORTED otherwise (nefes_blob_info refuses C > 141 and widths other than 128 / 256 up front).
 * Same arguments, outputs and ReLU-mask words as the calls above, so forward and backward kernels of every kind combine.
 * nefes_pack_device refreshes the fp16 streams when it is given the plan of nefes_pack_h3_plan. */
int nefes_field_fwd_h3(const NefesNetDesc* desc, const void* packed, int mode, int N, int S, const float* rays_o,
                       const float* rays_d, const float* z, const float* pts, const float* xyz_enc, const float* viewdirs,
                       float* raw_t, uint32_t* masks, void* stream);
/* nefes_field_fwd_h3 with ONE row of S depths shared by every ray (z_row [S], frequency embedding): the coarse pass at test time
 * with scalar near / far and no jitter (rendering.py:96-100: z_vals = near (1 - t) + far t, then expand) -- the [N, S] tensor is
 * never materialised. */
int nefes_field_fwd_h3_zrow(const NefesNetDesc* desc, const void* packed, int mode, int N, int S, const float* rays_o,
                            const float* rays_d, const float* z_row, const float* viewdirs, float* raw_t, uint32_t* masks,
                            void* stream);
int nefes_field_bwd_h3(const NefesNetDesc* desc, const void* packed, int N, int S, const float* rays_o, const float* rays_d,
                       const float* z, const float* pts, const float* viewdirs, const float* raw_t, const float* g_raw_t,
                       const uint32_t* masks, float* g_pts, float* g_xyz_enc, float* g_viewdirs_s, void* stream);
/* BASELINE configs[3] with the hash grid INSIDE the field kernels (round 5): for a NEFES_XYZ_EXTERNAL32 network whose 32 features
 * are the multiresolution hash grid of pts = o + d z (script/models/nerfh_tcnn.py:60-75,151-182; `grid` / `table` as for
 * nefes_hashgrid_fwd, sixteen levels), the forward gathers the table in its prologue and the backward returns g_pts [N*S, 3] =
 * d loss / d pts through the MLP and the grid (frozen table) -- nefes_hashgrid_fwd + nefes_field_fwd_h3(xyz_enc) and
 * nefes_field_bwd_h3(g_xyz_enc) + nefes_hashgrid_bwd_x without the two [M, 32] tensors (10 GB each per fine pass at 854x480).
 * Forward outputs are bit-identical to that sequence.  z: [N, S], or (forward, z_is_row != 0) one row [S] shared by every ray.
 * Width 256, C <= 29.  PARITY UNPINNED like nefes_hashgrid_fwd. */
int nefes_field_fwd_h3_hashgrid(const NefesNetDesc* desc, const void* packed, const NefesHashGridDesc* grid, const float* table,
                                int mode, int N, int S, const float* rays_o, const float* rays_d, const float* z, int z_is_row,
                                const float* viewdirs, float* raw_t, uint32_t* masks, void* stream);
int nefes_field_bwd_h3_hashgrid(const NefesNetDesc* desc, const void* packed, const NefesHashGridDesc* grid, const float* table,
                                int N, int S, const float* rays_o, const float* rays_d, const float* z, const float* viewdirs,
                                const float* raw_t, const float* g_raw_t, const uint32_t* masks, float* g_pts, float* g_viewdirs_s,
                                void* stream);
/* FACTORED HEAD (round 5).  The static rgb+feature head is linear in g = relu(dir_encoding output) (no activation when C > 0:
 * script/models/nerfh_nff.py:487-490) and so is compositing (feat = sum_s w_s (W_f g_s + b_f), :119-125), so for a network whose head
 * has more channels than g -- the reference's default: 3 + 128 channels against W/2 = 64 -- the field kernels can emit g instead of the
 * feature channels and the caller applies W_f once per RAY to the composited g.  `desc` / `packed`: the network packed WITHOUT its
 * feature rows (feat_dim = 0, static_rgb = its first three rows); raw_t / g_raw_t [N][3 + (W/2 + 1) + 6][S] = rgb (3) | g (W/2) | a
 * channel of ones (its composite is sum_s w_s, the bias's factor) | sigma | transient rgb (3), sigma, beta: what nefes_composite_fwd /
 * _bwd take with C = W/2 + 1.  Width 128, frequency embedding, full head, frozen weights. */
int nefes_field_fwd_h3_fh(const NefesNetDesc* desc, const void* packed, int mode, int N, int S, const float* rays_o,
                          const float* rays_d, const float* z, const float* viewdirs, float* raw_t, uint32_t* masks, void* stream);
/* g_gmap (nullable) [N][W/2 + 1] = d loss / d (composited g) per RAY: the gradient of a sample's g channels is then w_s g_gmap[ray][f], formed
 * in the kernel from the static weight w_s that nefes_composite_bwd(NEFES_COMP_FEAT_WEIGHTS_ONLY) leaves in the first feature channel's row of
 * g_raw_t -- the W/2 rows of products are neither written nor read.  Null: g_raw_t carries d loss / d g itself. */
int nefes_field_bwd_h3_fh(const NefesNetDesc* desc, const void* packed, int N, int S, const float* rays_o, const float* rays_d,
                          const float* z, const float* viewdirs, const float* raw_t, const float* g_raw_t, const float* g_gmap,
                          const uint32_t* masks, float* g_pts, float* g_viewdirs_s, void* stream);
/* the factored head's per-ray part: feat [N, C] = gmap[:, :F] W^T + gmap[:, F] b with gmap [N, F + 1] = the composited g and ones
 * channels (nefes_composite_fwd with C = F + 1), w_t = W transposed [F, C]; and its backward to gmap (w = W [C, F]; W, b frozen).
 * Every output is a sequential sum, independent of the batch: shards and batches stay bit-identical.  C <= 256, F < 256. */
int nefes_feat_head_fwd(int N, int C, int F, const float* gmap, const float* w_t, const float* b, float* feat, void* stream);
int nefes_feat_head_bwd(int N, int C, int F, const float* g_feat, const float* w, const float* b, float* g_gmap, void* stream);
/* nefes_field_bwd_static on the fp16 two-part pipe (round 5): backward-to-inputs of a NEFES_FIELD_STATIC forward
 * (nefes_field_fwd_h3 accepts that mode for the frequency embedding) for every compiled (width, head class) pair -- a frozen coarse
 * network with test_time False (script/models/rendering.py:116-125) or a fine network with NeRFW off
 * (script/models/nerfh_nff.py:217-231, output_transient False).  Same arguments as nefes_field_bwd_static. */
int nefes_field_bwd_static_h3(const NefesNetDesc* desc, const void* packed, int N, int S, const float* rays_o, const float* rays_d,
                              const float* z, const float* pts, const float* viewdirs, const float* raw_t, const float* g_raw_t,
                              const uint32_t* masks, float* g_pts, float* g_viewdirs_s, void* stream);

/* ---- train mode: weight gradients (script/run_nefes.py:42-108 `loss.backward()` through models/nerfh_nff.py:525-576) ----
 * Buffers `acts` / `dacts`: fp32 [n_tiles = ceil(N*S/128)][rows x 128 samples], rows = nefes_train_rows(desc); inside a tile
 * element (row, sample) sits at float offset [row / 32][sample / 16][row % 32][sample % 16] (csrc/layout.h nefes_train_off); row blocks
 * NEFES_TB_* (nefes_amd/csrc/layout.h): E, DV (embeddings, slot order), L1..L8, FINAL, DIR, T0..T2 (natural feature
 * order), RGB, SIG, TH (head gradients, padded to 32 rows).  `acts` holds PRE-activations, `dacts` their gradients. */
size_t nefes_train_rows(const NefesNetDesc* desc);
int nefes_train_row_offset(const NefesNetDesc* desc, int block);      /* block 0..18 (18 = rows per tile) */
/* nefes_field_fwd (mode STATIC or FULL, frequency embedding) that also writes `acts` (and the ReLU masks if non-null). */
int nefes_field_fwd_train(const NefesNetDesc* desc, const void* packed, int mode, int N, int S, const float* rays_o,
                          const float* rays_d, const float* z, const float* pts, const float* viewdirs, float* raw_t,
                          float* acts, uint32_t* masks, void* stream);
/* The same on the fp16 two-part pipe (nefes_field_fwd_h3 instances with the static-head / full streams; widths 256 / C = 16
 * and 128 / C = 128): same raw_t, acts rows and mask words. */
int nefes_field_fwd_train_h3(const NefesNetDesc* desc, const void* packed, int mode, int N, int S, const float* rays_o,
                             const float* rays_d, const float* z, const float* pts, const float* viewdirs, float* raw_t,
                             float* acts, uint32_t* masks, void* stream);
/* The fused backward kernel (nefes_field_bwd / _static) that also stores, for every hidden layer, the gradient w.r.t. its
 * pre-activation into `dacts` (blocks L1..L8, FINAL, DIR, T0..T2): replaces the nefes_train_dx chain.  g_pts / g_viewdirs_s
 * [N*S,3] receive the per-sample input gradients as in nefes_field_bwd. */
int nefes_field_bwd_train(const NefesNetDesc* desc, const void* packed, int mode, int N, int S, const float* rays_o,
                          const float* rays_d, const float* z, const float* viewdirs, const float* raw_t,
                          const float* g_raw_t, const uint32_t* masks, float* dacts, float* g_pts, float* g_viewdirs_s,
                          void* stream);
/* The same on the fp16 two-part pipe (nefes_field_bwd_h3 instances, static-head / full streams): same dacts rows. */
int nefes_field_bwd_train_h3(const NefesNetDesc* desc, const void* packed, int mode, int N, int S, const float* rays_o,
                             const float* rays_d, const float* z, const float* viewdirs, const float* raw_t,
                             const float* g_raw_t, const uint32_t* masks, float* dacts, float* g_pts, float* g_viewdirs_s,
                             void* stream);
/* d raw_t [N][R][S] -> head pre-activation gradients in dacts blocks RGB, SIG (, TH); samples beyond N*S are zeroed. */
int nefes_train_head_grad(const NefesNetDesc* desc, int mode, int N, int S, const float* raw_t, const float* g_raw_t,
                          float* dacts, void* stream);
/* dacts_out[dst_row0 + i][s] (+)= sum_{o < n_out} wt[i][o] * dacts_in[g_row0 + o][s],  i < n_in (64, 128 or 256);
 * wt = TRANSPOSED weights [n_in][ldw] (n_out % 8 == 0, zero padded); mask: multiply by [acts[dst_row0 + i][s] > 0]. */
int nefes_train_dx(int64_t n_tiles, int rows, const float* dacts_in, int g_row0, int n_out, const float* wt, int ldw,
                   int n_in, const float* acts, int dst_row0, int accumulate, int mask, float* dacts_out, void* stream);
/* partial[sp][o][i] = sum over the sp-th share of the sample tiles of dacts[g_row0 + o][s] * f(acts[x_row0 + i][s]),
 * f = ReLU if x_relu else identity; n_out, n_in multiples of 32; the caller sums the `splits` partials. */
int nefes_train_dw(int64_t n_tiles, int rows, const float* dacts, int g_row0, int n_out, const float* acts, int x_row0,
                   int n_in, int x_relu, int splits, float* partial, void* stream);
/* The same with the bias gradient riding along: partial[sp][o][n_in + 1], column n_in = sum over the share's samples of
 * dacts[g_row0 + o][s] (the row sums autograd gives a Linear's bias; the operand is in registers for the product anyway, a
 * separate reduction would read the whole gradient buffer a second time).  split_stride = floats between the partials of
 * consecutive shares (0: n_out * (n_in + 1), back to back) -- a caller that gives every layer's launch the same `splits` and one
 * wide buffer sums all layers' partials with a single reduction. */
int nefes_train_dw_bias(int64_t n_tiles, int rows, const float* dacts, int g_row0, int n_out, const float* acts, int x_row0,
                        int n_in, int x_relu, int splits, int64_t split_stride, float* partial, void* stream);

/* ---- FusionNet's convolutions (script/models/nerfh_nff.py:356-418; run on the rendered 60x80 image by run_fusion_net in
 *      every refinement iteration, script/dm/DFM_APR_refine.py:112-120) ----
 * y[b][co][p] = bias[co] + sum_{tap, ci} W[co][ci][tap] x[b][ci][p + tap - pad]  (stride 1, zero "same" padding, ksize 3 or 5;
 * ReLU on the way out if relu).  NCHW fp32.  w_packed = [Cin rounded up to even][ksize*ksize][Cout rounded up to 32] =
 * W[co][ci][ty][tx] at [ci][ty*ksize + tx][co], zero padded (the caller packs once per weight version).  mask (nullable,
 * x's shape): x is read as zero where mask <= 0 -- the ReLU derivative of the layer in front, for the gradient pass: the gradient
 * w.r.t. a layer's input is this call on the flipped, transposed weights (W'[ci][co][ty][tx] = W[co][ci][K-1-ty][K-1-tx]). */
int nefes_conv2d_same(int B, int Cin, int Cout, int H, int W, int ksize, const float* x, const float* mask, const float* w_packed,
                      const float* bias, int relu, float* y, void* stream);

/* ---- bicubic up-sampling of the fused feature image (script/dm/DFM_APR_refine.py:114,118: torch.nn.Upsample(size,
 *      mode='bicubic'), align_corners=False, A=-0.75) ---- */
/* in [planes,h,w] -> out [planes,CH,CW] = the window [oy0, oy0+CH) x [ox0, ox0+CW) of the OH x OW up-sampled image (planes =
 * batch*channels, contiguous NCHW).  The loop crops 10 pixels per side right after up-sampling (DFM_APR_refine.py:115,119): pass
 * the crop here and only the window is computed; the full image is (0, 0, OH, OW). */
int nefes_bicubic_up_fwd(int64_t planes, int h, int w, int OH, int OW, int oy0, int ox0, int CH, int CW, const float* in, float* out,
                         void* stream);
/* g_out [planes,CH,CW] (gradient of the window) -> g_in [planes,h,w]; separable gather (no atomics, deterministic);
 * tmp: [planes,h,CW] scratch. */
int nefes_bicubic_up_bwd(int64_t planes, int h, int w, int OH, int OW, int oy0, int ox0, int CH, int CW, const float* g_out, float* tmp,
                         float* g_in, void* stream);

/* ---- the per-image refinement loop's glue (script/dm/DFM_pose_refine.py:290-348) ---- */
/* LearnPose.forward (script/models/poses.py:43-50, lietorch=False: utils/lie_group_helper.py:60-81) + fix_coord_supp
 * (script/dm/direct_pose_model.py:224-231): c2w [3,4] = [Exp(r) R0 | ((t + t0) sc + move) sc2].
 * n_poses cameras at once: r, t: dev [n,3]; init_c2w: dev [n,4,4] row-major; move: HOST [3]; c2w: dev [n,3,4]. */
int nefes_pose_compose_fwd(int n_poses, const float* r, const float* t, const float* init_c2w, float pose_scale, const float* move,
                           float pose_scale2, float* c2w, void* stream);
/* g_c2w dev [n,3,4] -> g_r, g_t dev [n,3] (analytic derivative of the Rodrigues formula, float64 inside). */
int nefes_pose_compose_bwd(int n_poses, const float* r, const float* t, const float* init_c2w, float pose_scale, const float* move,
                           float pose_scale2, const float* g_c2w, float* g_r, float* g_t, void* stream);
/* FusionNet's BatchNorm2d in TRAIN mode with frozen affine parameters (nerfh_nff.py:356-418; torch.nn.BatchNorm2d semantics): x dev
 * [B,C,P] -> y, normalised by the batch's statistics (per_image = 0; running_mean / running_var dev [C] updated with `momentum`, the int64 batch counter incremented; any of them NULL: left alone)
 * or by every image's own (per_image = 1: what the reference's one-image-at-a-time loop computes for a batch; running statistics
 * untouched).  float64 sums.  save: dev [groups * C * 2] doubles (mean, 1 / sqrt(var + eps)) for the backward, which returns d x only. */
int nefes_bn_train_fwd(int B, int C, int64_t P, int per_image, const float* x, const float* weight, const float* bias, double eps,
                       double momentum, float* running_mean, float* running_var, int64_t* num_batches_tracked, float* y, double* save,
                       void* stream);
int nefes_bn_train_bwd(int B, int C, int64_t P, int per_image, const float* x, const float* weight, const double* save, const float* g_y,
                       float* g_x, void* stream);
/* svd_reg (dm/DFM_pose_refine.py:119-129): pose dev [n,3,4] -> out dev [n,3,4] with the 3x3 block replaced by U V^T of its SVD (its
 * orthogonal polar factor), translation column copied.  float64 inside (one-sided Jacobi).  save: dev [n,21] doubles (U, V, sigma) for the
 * backward, or NULL.  The backward is the polar factor's derivative d A = U [(H - H^T) o K] V^T, H = U^T G V, K_ij = 1 / (s_i + s_j):
 * well conditioned at the near-rotations a pose network regresses, where autograd through torch.svd divides by s_i^2 - s_j^2 ~ 0. */
int nefes_svd_reg_fwd(int n_poses, const float* pose, float* out, double* save, void* stream);
int nefes_svd_reg_bwd(int n_poses, const double* save, const float* g_out, float* g_pose, void* stream);
/* The pose chain behind the regression network of train_on_batch (dm/DFM_APR_refine.py:91-97) as one launch each way: svd_reg as above
 * (do_svd = 0: the rotation block is copied) and fix_coord_supp's translation (dm/direct_pose_model.py:210-232: t' = (t sc + move) sc2,
 * passed as t_scale = sc sc2 and (mx, my, mz) = move sc2; fp32, one multiply and one add as the torch expression rounds).  The backward
 * returns d pose [n,3,4]: the polar factor's derivative for the block, g_t t_scale for the translation. */
int nefes_regressed_pose_fwd(int n_poses, const float* pose, int do_svd, float t_scale, float mx, float my, float mz, float* out, double* save,
                             void* stream);
int nefes_regressed_pose_bwd(int n_poses, const double* save, int do_svd, float t_scale, const float* g_out, float* g_pose, void* stream);
/* The verification step of train_on_batch (DFM_APR_refine.py:117-128, :146-150): out[0] = mse2psnr(img2mse(x, y)) (models/nerfh.py),
 * out[1] = SSIM()(x, y).mean() (utils/utils.py:15-49: ReflectionPad2d(3), AvgPool2d(7, 1), clamp(n / d, 0, 1)) of two images of C planes
 * H x W, each with a row and a plane stride in elements (the loop's 10-pixel crop is a view).  workspace: nefes_psnr_ssim_workspace bytes.
 * Two launches instead of torch's ~25; fp32 window sums as avg_pool2d, float64 sums over the image. */
size_t nefes_psnr_ssim_workspace(int C, int H, int W);
int nefes_psnr_ssim(int C, int H, int W, const float* x, int64_t x_row_stride, int64_t x_plane_stride, const float* y, int64_t y_row_stride,
                    int64_t y_plane_stride, void* workspace, float* out, void* stream);
/* feature_loss (DFM_pose_refine.py:211-233, per_pixel=False): loss = 1 - mean_c cos(a[c,:], b[c,:]), a, b dev [C,P] contiguous,
 * torch.nn.CosineSimilarity(dim=1, eps=1e-6) semantics, float64 accumulation.  scratch: dev doubles,
 * nefes_cosine_loss_scratch_doubles(C) of them, kept by the caller for the backward. */
size_t nefes_cosine_loss_scratch_doubles(int C);
int nefes_cosine_loss_fwd(int C, int64_t P, const float* a, const float* b, double* scratch, float* loss, void* stream);
/* g_a [C,P] = g_loss[0] * d loss / d a (g_loss: dev scalar). */
int nefes_cosine_loss_bwd(int C, int64_t P, const float* a, const float* b, const double* scratch, const float* g_loss, float* g_a,
                          void* stream);

/* feature_loss ON the bicubically up-sampled, cropped feature image without materialising it (DFM_APR_refine.py:114,125-131):
 * x [C,h,w] is up-sampled to (OH, OW) as nefes_bicubic_up_fwd does, `crop` pixels are dropped on every side, and the loss is
 * nefes_cosine_loss_fwd against target [C, OH-2crop, OW-2crop] -- each up-sampled value interpolated where it is consumed.
 * scratch: nefes_cosine_loss_scratch_doubles(C) doubles, kept for the backward; tmp: [C, OH-2crop, w] floats. */
int nefes_upcos_loss_fwd(int C, int h, int w, int OH, int OW, int crop, const float* x, const float* target, double* scratch, float* loss,
                         void* stream);
/* tx_* / ty_*: gather tables of the x and y axes (nefes_bicubic_gather_table with (w, OW, crop, OW-2crop) and (h, OH, crop, OH-2crop)),
 * BOTH built with the row length T passed here (the larger of the two axes' needs when OH / h != OW / w). */
int nefes_upcos_loss_bwd(int C, int h, int w, int OH, int OW, int crop, const float* x, const float* target, const double* scratch,
                         const float* g_loss, const int* tx_first, const int* tx_count, const float* tx_wt, const int* ty_first,
                         const int* ty_count, const float* ty_wt, int T, float* tmp, float* g_x, void* stream);
/* The same loss for a FIXED target (the refinement loop holds the query image's features for all opt_iter iterations of an image:
 * DFM_APR_refine.py:100-131), through the Gram matrices of the up-sampling.  Per channel up = Uy X Ux^T, so
 *     <up, target> = <X, Tt>, Tt = Uy^T target Ux;   |up|^2 = <X, Gy X Gx>, Gy = Uy^T Uy, Gx = Ux^T Ux;   |target|^2 a constant,
 * and an iteration reads X, Tt and writes P = Gy X Gx (each [C,h,w]) instead of the [C, OH-2crop, OW-2crop] target twice.  float64 sums
 * over the float32 tap weights of nefes_bicubic_up_fwd.
 *   nefes_bicubic_gram:   G dev [n_in, n_in] doubles of one axis' window (once per geometry; zero outside |a - b| <= 3).
 *   nefes_upcos_prepare:  once per target: tt dev [C,h,w] doubles, dbb dev [C] doubles; tmp dev [C, OH-2crop, w] doubles (work space);
 *                         tx_* / ty_* the gather tables of nefes_upcos_loss_bwd.
 *   nefes_upcos_gram_fwd: loss and scratch as nefes_upcos_loss_fwd writes them (same layout, same cosine_final launch);
 *                         pmat dev [C,h,w] doubles kept for the backward; band >= the Gram matrices' half band width (3).
 *   nefes_upcos_gram_bwd: g_x [C,h,w] = g_loss[0] * d loss / d x. */
int nefes_bicubic_gram(int n_in, int n_out, int o0, int n_win, double* G, void* stream);
int nefes_upcos_prepare(int C, int h, int w, int OH, int OW, int crop, const float* target, const int* tx_first, const int* tx_count,
                        const float* tx_wt, const int* ty_first, const int* ty_count, const float* ty_wt, int T, double* tmp, double* tt,
                        double* dbb, void* stream);
int nefes_upcos_gram_fwd(int C, int h, int w, const float* x, const double* tt, const double* dbb, const double* gram_x, const double* gram_y,
                         int band, double* scratch, double* pmat, float* loss, void* stream);
/* LDS bytes one workgroup of nefes_upcos_gram_fwd needs at this geometry; above 96 KiB the launch returns NEFES_E_UNSUPPORTED and the
 * caller stays on the one-pass kernels (nefes_upcos_loss_fwd / _bwd), which take any geometry (asked when the target is prepared). */
size_t nefes_upcos_gram_lds_bytes(int h, int w, int band);
int nefes_upcos_gram_bwd(int C, int h, int w, const double* tt, const double* pmat, const double* scratch, const float* g_loss, float* g_x,
                         void* stream);
/* Which up-sampled positions of the window [o0, o0 + n_win) of an axis (n_in -> n_out, bicubic) reach source index y, and with what
 * weight: first[y] (relative to o0), count[y], wt[y * T .. + count[y]) -- the transpose of the interpolation, a function of the sizes
 * only; T >= 4 n_out / n_in + 8. */
int nefes_bicubic_gather_table(int n_in, int n_out, int o0, int n_win, int T, int* first, int* count, float* wt, void* stream);

/* ---- FusionNet's input from the rendered maps, one launch each way (script/models/nerfh_nff.py:605-626 affine_color_transform
 *      with the image's 12 exposure coefficients given, :578-603 run_fusion_net's reshape / permute / cat, :395-402 the colour
 *      normalisation):  y = sigmoid(K rgb + b) (affine == NULL: y = rgb);  x[b, 0:3] = (y - mean) / std;  x[b, 3:] = feat^T.
 *      rgb [B*HW,3], feat [B*HW,C], affine [B,12] (dev), mean3 / std3 HOST pointers -> x [B,3+C,HW], y [B*HW,3] (saved). ---- */
int nefes_fusion_input_fwd(int B, int HW, int C, const float* rgb, const float* feat, const float* affine, const float* mean3,
                           const float* std3, float* x, float* y, void* stream);
/* g_x [B,3+C,HW] -> g_rgb [B*HW,3], g_feat [B*HW,C] (either may be NULL); the exposure coefficients carry no gradient here. */
int nefes_fusion_input_bwd(int B, int HW, int C, const float* g_x, const float* affine, const float* y, const float* std3, float* g_rgb,
                           float* g_feat, void* stream);
/* torch.optim.Adam's update (no weight decay, no amsgrad) for n <= 1024 fp32 numbers with a learning rate per element (dev array
 * of doubles; bias corrections and step size in float64 like the Python scalars of torch/optim/adam.py): the refinement loop's
 * (r, t) with (lr_r, lr_t) in one launch.  step: dev scalar (float), incremented. */
int nefes_adam_step(int n, float* p, const float* g, float* m, float* v, float* step, const double* lr, double beta1, double beta2,
                    double eps, void* stream);

/* ---- measurement aid (bench.py's roofline): the matrix-core rate this GPU SUSTAINS under its power management.  Runs
 *      v_mfma_f32_32x32x16_f16 back to back on every SIMD for ~ms_target milliseconds (operands all zero, or random bits) and
 *      returns the settled shader clock and the dense fp16 rate.  Synchronises the stream.  Not part of the render path. ---- */
int nefes_probe_mfma_clock(int random_operands, int ms_target, double* clock_ghz, double* fp16_dense_tflops, void* stream);
/* Diagnostic (tools/store_hazard.py): out dev [8][n_float4][4] floats; lane i stores (v, v, v + j, v), v = (i mod 2^20) + 1, to plane
 * j = 0..7 with global_store_dwordx4 from the SAME four registers, adding 1 to the third `nops` + 1 wait states behind each store
 * (nops in {0, 1, 3, 7, 15}).  out[j][i][2] != v + j = that store went out with a later value of its register. */
int nefes_probe_store_hazard(float* out, int64_t n_float4, int nops, void* stream);
/* Diagnostic: every lane runs `iters` v_pk_mul_f32 with op_sel:[0,1] / op_sel_hi:[1,0] on fresh operands; out dev [n] = how many of its
 * results were not the two products. */
int nefes_probe_pk_mul(unsigned* out, int64_t n, int iters, void* stream);
/* What the hardware itself does with the producer -> consumer pairs tools/hazard_lint.py checks (csrc/hazard_probe.hip):
 * every SIMD runs `iters` repetitions of one pair with K wait states in between; *count (device, zeroed by the caller) += the lanes x
 * repetitions whose result differs from the result with the full distance.  test: 0 fp32 MFMA result -> vector read, 1 / 2 16-bit MFMA
 * result -> v_mov / v_accvgpr_read, 3 / 4 MFMA read of SrcB / SrcC -> vector overwrite, 5 / 6 vector write of SrcB / SrcC -> MFMA,
 * 7 v_cmp VCC -> v_cndmask, 8 MFMA result -> next MFMA's SrcB, 9 MFMA result -> vector overwrite, 10 MFMA result -> ds_write,
 * 11 vector write -> v_permlane32_swap, 12 v_exp -> vector read, 13 vector write -> DPP read, 14 vector write -> v_readfirstlane,
 * 15 v_accvgpr_write of SrcC -> MFMA;
 * K in {0 .. 8, 10, 12, 16, 18}. */
int nefes_probe_hazard(int test, int k, int blocks, int iters, unsigned* count, void* stream);
/* Diagnostic neighbour for nefes_probe_pk_mul: `blocks` workgroups of 256 run `iters` rounds of one instruction kind (0: v_fma_mixlo/hi_f16,
 * 1: v_mfma_f32_32x32x16_f16, 2: both, 3: v_pk_fma_f32 with op_sel_hi, 4: v_fma_f32); sink dev [blocks * 256] floats. */
int nefes_probe_aggressor(int kind, int iters, int blocks, float* sink, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* NEFES_HIP_H */
