"""Field network: host-side mirror of the reference's NeRFH_NFF module and its query function.

Mirrors script/models/nerfh_nff.py (reference): class NeRFH_NFF (:421-626), FusionNet (:356-418),
get_embedder/Embedder (:234-354), run_network_NeRFH_NFF (:168-231).  Parameter names and
construction order are identical, so reference checkpoints load with load_state_dict and
`torch.manual_seed(0)` random init is bit-identical (checked in tests against stored checksums).

The per-sample evaluation on the render path is NOT torch: `run_network_NeRFH_NFF` and
`render()` hand the module to the fused HIP kernels through `module.packed()`.
`NeRFH_NFF.forward` (pre-embedded inputs, the nn.Module API) is kept in torch only so that code
which calls the module directly keeps working; nothing on the render path calls it.
"""
import os

import torch
import torch.nn as nn

from . import lib as L
from . import ops

FEATURE_DIM = 128        # nerfh_nff.py:21


class FusionNet(nn.Module):
    """4-layer conv 'fusion' CNN on the rendered (rgb || feature) image (nerfh_nff.py:356-418).
    Plain torch (MIOpen) where its weights train; in the refinement loop, where they are frozen and the net runs
    on the 60x80 render every iteration, forward_parts takes the implicit-GEMM kernels of csrc/conv.hip."""
    mean = [0.485, 0.456, 0.406]
    std = [0.229, 0.224, 0.225]

    def __init__(self, f_dim, fusion_residule=False, no_BN=False):
        super().__init__()
        self.fusion_residule, self.no_BN = fusion_residule, no_BN
        layers = [nn.Conv2d(3 + f_dim, 64, 3, 1, 1), nn.ReLU(), nn.Conv2d(64, 64, 3, 1, 1), nn.ReLU(),
                  nn.Conv2d(64, 64, 3, 1, 1), nn.ReLU(), nn.Conv2d(64, f_dim, 5, 1, 2)]
        if not no_BN:
            layers.append(nn.BatchNorm2d(f_dim))
        self.net = nn.Sequential(*layers)
        self._norm = {}                                             # (device, dtype) -> (mean, std); not in the state_dict

    def _mean_std(self, x):
        key = (x.device, x.dtype)
        if key not in self._norm:                                   # one host->device copy per device, outside any graph capture
            self._norm[key] = (x.new_tensor(self.mean), x.new_tensor(self.std))
        return self._norm[key]

    def forward(self, x):
        mean, std = self._mean_std(x)
        x[:, :3] = (x[:, :3] - mean[:, None, None]) / std[:, None, None]     # in place, as the reference does
        out = self.net(x)
        return x[:, 3:] + out if self.fusion_residule else out

    HIP_CONVS = True      # frozen weights on a GPU: the four convolutions as csrc/conv.hip launches (ops.frozen_conv2d)

    def _use_hip(self, x):
        """The refinement loop's case: CUDA input, no convolution parameter requires grad (training them goes through torch)."""
        return (self.HIP_CONVS and x.is_cuda and x.dtype == torch.float32
                and not any(p.requires_grad for m in self.net[:7] for p in m.parameters()))

    def _convs_hip(self, x):
        from . import ops
        c0, c1, c2, c3 = self.net[0], self.net[2], self.net[4], self.net[6]
        x = ops.frozen_conv2d(x, c0.weight, c0.bias, relu=True)
        x = ops.frozen_conv2d(x, c1.weight, c1.bias, relu=True)
        x = ops.frozen_conv2d(x, c2.weight, c2.bias, relu=True)
        return ops.frozen_conv2d(x, c3.weight, c3.bias, relu=False)

    HIP_BATCHNORM = os.environ.get("NEFES_HIP_BATCHNORM", "1") != "0"   # the refinement loop's case (train mode, frozen affine parameters, GPU): csrc/refine.hip bn_train_* instead of MIOpen

    # False: train-mode BatchNorm calls leave the module's running statistics and batch counter alone (its outputs never depend on
    # them; the reference's loop never reads them either: the net is never .eval()ed).  PoseRefiner(bn_running_stats=False) sets it
    # around its own calls: refiners on different streams share this module, and the update is a read-modify-write of its buffers.
    track_bn_stats = True

    def _bn(self, y, per_image_norm):
        """The last layer on the convolutions' output y [B,C,H,W]."""
        from . import ops
        bn = self.net[-1]
        per_image = bool(per_image_norm and y.shape[0] > 1 and bn.training)
        if (self.HIP_BATCHNORM and self.HIP_CONVS and y.is_cuda and y.dtype == torch.float32 and bn.training and bn.momentum is not None
                and not any(p is not None and p.requires_grad for p in (bn.weight, bn.bias))):
            return ops.batch_norm_train_frozen(y, bn, per_image, track_stats=self.track_bn_stats)
        if per_image:
            return nn.functional.instance_norm(y, weight=bn.weight, bias=bn.bias, eps=bn.eps)
        if bn.training and not self.track_bn_stats:
            return nn.functional.batch_norm(y, None, None, bn.weight, bn.bias, True, 0.0, bn.eps)
        return bn(y)

    def _conv0_on_gmap(self, w_f, b_f):
        """Conv2d(3 + C, 64, 3) o the factored feature head: the first layer's weights acting on (rgb, sum_s w_s g_s, sum_s w_s)
        [3 + Cg + 1 channels] instead of on (rgb, feat) -- feat = W_f gmap + b_f (sum_s w_s) per pixel (ops.RenderFineFH) and the layer is
        linear in feat; zero padding commutes (gmap = 0 outside the image gives feat = 0 there).  Composed once in float64 for frozen
        weights: [64, 3 + Cg + 1, 3, 3] fp32."""
        c0 = self.net[0]
        key = (c0.weight.data_ptr(), c0.weight._version, w_f.data_ptr(), w_f._version, b_f.data_ptr(), b_f._version)
        if getattr(self, "_gmap_conv0", None) is None or self._gmap_conv0[0] != key:
            with torch.no_grad():
                w1, wf, bf = c0.weight.detach().double(), w_f.detach().double(), b_f.detach().double()
                wg = torch.einsum("ocyx,cj->ojyx", w1[:, 3:], wf)
                wb = torch.einsum("ocyx,c->oyx", w1[:, 3:], bf)[:, None]
                self._gmap_conv0 = (key, torch.cat([w1[:, :3], wg, wb], 1).float().contiguous())
        return self._gmap_conv0[1]

    def forward_prepared_gmap(self, x, w_f, b_f, per_image_norm=False):
        """forward_prepared for x = [B, 3 + Cg + 1, H, W] holding the factored head's per-pixel INPUT in the features' place
        (render(..., feat_as_gmap=True)): the head is folded into the first convolution (_conv0_on_gmap), so neither the per-ray head
        kernels nor 128 feature channels exist.  Frozen weights on the GPU, no residual connection (it needs the features)."""
        from . import ops
        if self.fusion_residule or not self._use_hip(x):
            raise RuntimeError("nefes_amd: forward_prepared_gmap needs frozen FusionNet weights on the GPU and no residual connection")
        c0, c1, c2, c3 = self.net[0], self.net[2], self.net[4], self.net[6]
        y = ops.frozen_conv2d(x, self._conv0_on_gmap(w_f, b_f), c0.bias, relu=True)
        y = ops.frozen_conv2d(y, c1.weight, c1.bias, relu=True)
        y = ops.frozen_conv2d(y, c2.weight, c2.bias, relu=True)
        y = ops.frozen_conv2d(y, c3.weight, c3.bias, relu=False)
        return y if self.no_BN else self._bn(y, per_image_norm)

    def forward_parts(self, rgb_nchw, feat_nchw, per_image_norm=False):
        """forward(cat([rgb, feat], 1)) without the in-place slice assignment (whose autograd costs a fill and two copies):
        the colour channels are normalised before the concatenation -- same values, same order of operations.
        per_image_norm: B independent images in one batch (PoseRefiner(images=B)).  The reference runs this net on one image at
        a time with BatchNorm in train mode, i.e. normalised by that image's own statistics; for a batch the same arithmetic is
        an instance norm with the BatchNorm's affine parameters (the running statistics, which train mode never reads, are
        not updated on this path).  A BatchNorm put into eval() normalises every image by its running statistics instead --
        per image already -- so the batch then goes through the module itself, like a single image does."""
        mean, std = self._mean_std(rgb_nchw)
        x = torch.cat([(rgb_nchw - mean[:, None, None]) / std[:, None, None], feat_nchw], dim=1)
        return self.forward_prepared(x, per_image_norm)

    def forward_prepared(self, x, per_image_norm=False):
        """forward_parts from the concatenated, colour-normalised input [B,3+C,H,W] on (ops.fusion_input builds it in one launch)."""
        feat_nchw = x[:, 3:]
        convs = self._convs_hip if self._use_hip(x) else self.net[:7]
        out = convs(x) if self.no_BN else self._bn(convs(x), per_image_norm)
        return feat_nchw + out if self.fusion_residule else out


class ExposureMLP(nn.Module):
    """tcnn-free stand-in for tcnn.Network(10 -> 12, FullyFusedMLP, 32 neurons, 3 hidden layers, ReLU, no bias)
    (nerfh_nff.py:511-522).  Keeps the flat `params` vector of the checkpoints: matrices stored back to back,
    each [out, in] row-major with the input padded 10->16 and the output padded 12->16 (tiny-cuda-nn's published
    FullyFusedMLP layout; parity UNPINNED -- tiny-cuda-nn is not vendored by the reference)."""
    SHAPES = [(32, 16), (32, 32), (32, 32), (16, 32)]

    def __init__(self, n_input_dims=10, n_output_dims=12):
        super().__init__()
        self.n_in, self.n_out = n_input_dims, n_output_dims
        n = sum(o * i for o, i in self.SHAPES)
        self.params = nn.Parameter(torch.empty(n).uniform_(-0.2, 0.2))

    def forward(self, x):
        h = torch.nn.functional.pad(x.float(), (0, 16 - self.n_in))
        off = 0
        for k, (o, i) in enumerate(self.SHAPES):
            w = self.params[off:off + o * i].view(o, i)
            off += o * i
            h = h @ w.t()
            if k + 1 < len(self.SHAPES):
                h = torch.relu(h)
        return h[:, :self.n_out]


class NeRFH_NFF(nn.Module):
    """Same constructor, attributes and parameter names as the reference (nerfh_nff.py:421-522)."""

    def __init__(self, typ, D=8, W=256, skips=[4], in_channels_xyz=63, in_channels_dir=27, encode_appearance=False,
                 in_channels_a=48, encode_transient=False, in_channels_t=16, beta_min=0.1, out_ch_size=3,
                 f_dim=FEATURE_DIM, fusion_residule=False, no_BN=False):
        super().__init__()
        torch.manual_seed(0)                                         # reference side effect (:446)
        self.typ, self.D, self.W, self.skips = typ, D, W, list(skips)
        self.in_channels_xyz, self.in_channels_dir = in_channels_xyz, in_channels_dir
        self.encode_appearance = False if typ == 'coarse' else encode_appearance
        self.encode_transient = False if typ == 'coarse' else encode_transient
        self.beta_min = beta_min
        self.W_features = f_dim
        self.out_ch_size = out_ch_size + f_dim
        self.fusion_residule, self.no_BN = fusion_residule, no_BN
        for i in range(D):
            n_in = in_channels_xyz if i == 0 else (W + in_channels_xyz if i in self.skips else W)
            setattr(self, f"xyz_encoding_{i + 1}", nn.Sequential(nn.Linear(n_in, W), nn.ReLU(True)))
        self.xyz_encoding_final = nn.Linear(W, W)
        self.dir_encoding = nn.Sequential(nn.Linear(W + in_channels_dir, W // 2), nn.ReLU(True))
        self.static_sigma = nn.Sequential(nn.Linear(W, 1), nn.Softplus())
        if self.out_ch_size == 3:
            self.static_rgb = nn.Sequential(nn.Linear(W // 2, 3), nn.Sigmoid())
        else:
            self.static_rgb = nn.Sequential(nn.Linear(W // 2, self.out_ch_size))
        if self.encode_transient:
            self.transient_encoding = nn.Sequential(nn.Linear(W + in_channels_dir, W // 2), nn.ReLU(True),
                                                    nn.Linear(W // 2, W // 2), nn.ReLU(True),
                                                    nn.Linear(W // 2, W // 2), nn.ReLU(True))
            self.transient_sigma = nn.Sequential(nn.Linear(W // 2, 1), nn.Softplus())
            if out_ch_size == 3:
                self.transient_rgb = nn.Sequential(nn.Linear(W // 2, 3), nn.Sigmoid())
            else:
                self.transient_rgb = nn.Sequential(nn.Linear(W // 2, self.out_ch_size))
            self.transient_beta = nn.Sequential(nn.Linear(W // 2, 1), nn.Softplus())
        if typ == 'coarse':
            self.fusion_net = FusionNet(self.W_features, fusion_residule, no_BN)
            self.exposure_embedding = ExposureMLP(10, 12)
            self.sigmoid = nn.Sigmoid()
        self._pk = None
        self._pk_key = None

    # -- the HIP path ---------------------------------------------------------------------------
    def _supported(self):
        # Encodings: the kernels compute the 10 / 4-frequency embeddings (63 / 27 features).  A network built on FEWER frequencies
        # -- `multires` < 10, `multires_views` < 4, or the reference's reduce_embedding modes 0 (half the frequencies) and 1 (none:
        # the raw 3-vector), nerfh_nff.py:307-330 -- reads a PREFIX of those features (x, then sin / cos per frequency in ascending
        # order), so it runs on the same kernels with zero weight columns for the frequencies it does not have (_kernel_params).
        freq_xyz = self.in_channels_xyz in range(3, 64, 6)
        return (self.D == 8 and self.skips == [4] and (freq_xyz or self.in_channels_xyz == 32) and self.in_channels_dir in range(3, 28, 6)
                and self.W in (128, 256) and self.out_ch_size != 3 and 0 < self.W_features <= ops.HEAD_MAX_C
                and (freq_xyz or (self.W == 256 and ops.head_class(self.W_features) == 0)))

    # layers whose input holds an embedding: (name, column where the embedding starts, its width here, its width in the kernels)
    def _embedding_columns(self):
        ex, ed, W = self.in_channels_xyz, self.in_channels_dir, self.W
        cols = []
        if ex != 32 and ex < 63:
            cols += [("xyz_encoding_1.0.weight", 0, ex, 63), ("xyz_encoding_5.0.weight", 0, ex, 63)]
        if ed < 27:
            cols += [("dir_encoding.0.weight", W, ed, 27)]
            if self.encode_transient:
                cols += [("transient_encoding.0.weight", W, ed, 27)]
        return cols

    def _kernel_params(self, sd):
        """Parameters in the shapes the kernels are built for: zero columns appended to a shorter embedding's weight block."""
        cols = self._embedding_columns()
        if not cols:
            return sd
        out = dict(sd)
        for name, at, have, want in cols:
            w = sd[name]
            out[name] = torch.cat([w[:, :at + have], w.new_zeros(w.shape[0], want - have), w[:, at + have:]], 1)
        return out

    def shrink_grads(self, g):
        """Inverse of _kernel_params on a {name: gradient} dict of the train-mode kernels (drops the padded columns)."""
        for name, at, have, want in self._embedding_columns():
            if name in g:
                g[name] = torch.cat([g[name][:, :at + have], g[name][:, at + want:]], 1)
        return g

    def invalidate_packed(self):
        """Force a re-pack on the next render.  The cache key is (data_ptr, _version, device) per parameter, which sees
        optimizer steps, load_state_dict and no_grad in-place ops on the parameter itself -- but NOT edits made through
        `.data` (`p.data.mul_()`, `p.data.copy_()`, EMA / clipping code): those leave both unchanged.  Call this after
        such an edit (a captured PoseRefiner graph must be re-captured as well)."""
        self._pk_key = None

    def packed(self) -> ops.PackedField:
        """Fragment streams for the fused kernels; re-packed when any path parameter changes (see invalidate_packed for
        the one kind of change this cannot see)."""
        if not self._supported():
            raise RuntimeError(f"nefes_amd: the HIP field kernels are built for D=8, skips=[4], frequency encodings of up to 10 / 4 "
                               f"octaves (3 + 6k inputs, k <= 10 / 4) and a feature head (f_dim>0); got D={self.D}, skips={self.skips}, "
                               f"W={self.W}, f_dim={self.W_features}, in_channels_xyz={self.in_channels_xyz}, "
                               f"in_channels_dir={self.in_channels_dir}.  Compiled: {ops.COMPILED_SET}")
        names = ops.PackedField.LAYERS_FINE if self.encode_transient else ops.PackedField.LAYERS_COARSE
        sd = dict(self.named_parameters())
        prm = [sd[n + s] for n in names for s in (".weight", ".bias")]
        key = tuple((p.data_ptr(), p._version, str(p.device)) for p in prm)
        pad = bool(self._embedding_columns())
        if os.environ.get("NEFES_DEBUG_PACK_CHECKSUM", "0") == "1":      # debug: also key on the values (one sync per call)
            key += (float(sum(p.detach().double().sum() for p in prm)),)
        if (self._pk is not None and key != self._pk_key and any(p.requires_grad for p in prm)
                and all(p.is_cuda and p.device == self._pk.blob.device for p in prm)):
            # a trainable network after an optimizer step: re-packed on the device, no host copy, no sync -- every stream, the
            # fp16 two-part ones and their scale tables included (ops.REPACK_H3; bit-identical to the host packer).  A FROZEN
            # network whose values changed (load_state_dict, an in-place edit) takes the host packer below.
            if pad:
                ksd = self._kernel_params(sd)
                self._pk.repack([ksd[n + s_] for n in names for s_ in (".weight", ".bias")])
            else:
                self._pk.repack(prm)
            self._pk_key = key
        elif self._pk is None or key != self._pk_key:
            dev = prm[0].device if prm[0].is_cuda else torch.device("cuda")
            enc = L.XYZ_EXTERNAL32 if self.in_channels_xyz == 32 else L.XYZ_FREQ10      # 32 = externally encoded (hash grid)
            self._pk = ops.PackedField(self._kernel_params({n: p.detach() for n, p in sd.items()}), self.W, self.W_features,
                                       self.encode_transient, dev, enc)
            self._pk_key = key
        return self._pk

    def factored_head_ok(self):
        """The factored-head kernels apply (csrc/field_fwd_h3.hip FH): a FROZEN fine network of width 128 on the frequency embedding whose
        rgb+feature head has more channels than g = relu(dir_encoding) has features (+ the ones channel), on the fp16 two-part pipe."""
        return (ops.FACTORED_HEAD and self.encode_transient and self.W == 128 and self.in_channels_xyz != 32 and self._supported()
                and 3 + self.W_features > 3 + self.W // 2 + 1 and ops.SPLIT == "h3"
                and not any(p.requires_grad for n, p in self.named_parameters()
                            if not n.startswith(("fusion_net", "exposure_embedding"))))

    def packed_fh(self):
        """(PackedField of the network WITHOUT its feature rows -- static_rgb = its three colour rows, f_dim 0 --, W_f [C, W/2], W_f^T, b_f [C]):
        what the factored-head kernels and the per-ray feature head of nefes_amd/render.py take.  Cached like packed()."""
        names = ops.PackedField.LAYERS_FINE
        sd = dict(self.named_parameters())
        prm = [sd[n + s] for n in names for s in (".weight", ".bias")]
        key = tuple((p.data_ptr(), p._version, str(p.device)) for p in prm)
        if getattr(self, "_pk_fh", None) is None or key != self._pk_fh_key:
            dev = prm[0].device if prm[0].is_cuda else torch.device("cuda")
            ksd = dict(self._kernel_params({n: p.detach() for n, p in sd.items()}))
            w, b = ksd["static_rgb.0.weight"], ksd["static_rgb.0.bias"]
            ksd["static_rgb.0.weight"], ksd["static_rgb.0.bias"] = w[:3].contiguous(), b[:3].contiguous()
            pk = ops.PackedField(ksd, self.W, 0, True, dev, L.XYZ_FREQ10)
            w_f = w[3:].to(dev, torch.float32).contiguous()
            self._pk_fh = (pk, w_f, w_f.t().contiguous(), b[3:].to(dev, torch.float32).contiguous())
            self._pk_fh_key = key
        return self._pk_fh

    # -- nn.Module API on pre-embedded inputs (not on the render path) ---------------------------
    def forward(self, x, sigma_only=False, output_transient=True):
        if sigma_only:
            input_xyz = x
        else:
            input_xyz, input_dir_a = torch.split(x, [self.in_channels_xyz, self.in_channels_dir], dim=-1)
        h = input_xyz
        for i in range(self.D):
            if i in self.skips:
                h = torch.cat([input_xyz, h], 1)
            h = getattr(self, f"xyz_encoding_{i + 1}")(h)
        static_sigma = self.static_sigma(h)
        if sigma_only:
            return static_sigma
        final = self.xyz_encoding_final(h)
        head_in = torch.cat([final, input_dir_a], 1)
        static = torch.cat([self.static_rgb(self.dir_encoding(head_in)), static_sigma], 1)
        if not output_transient:
            return static
        t = self.transient_encoding(head_in)
        return torch.cat([static, self.transient_rgb(t), self.transient_sigma(t), self.transient_beta(t)], 1)

    # -- post-render helpers used by the refinement loop (nerfh_nff.py:578-626) ----------------------
    def run_fusion_net(self, rgb, feature, H, W, B, per_image_norm=False):
        render_rgb = rgb.reshape(B, H, W, 3).permute(0, 3, 1, 2)
        render_feature = feature.reshape(B, H, W, self.W_features).permute(0, 3, 1, 2)
        fused = self.fusion_net.forward_parts(render_rgb, render_feature, per_image_norm)
        return render_rgb, render_feature, fused

    def exposure_coefficients(self, hist):
        """[B,10] histogram -> [B,12] (3x3 kernel, 3 bias): the first line of affine_color_transform (nerfh_nff.py:616)."""
        return self.exposure_embedding(hist.long()).float()

    def apply_affine(self, a_embedded, rgb, batch_size):
        """nerfh_nff.py:617-625: rgb' = sigmoid(K rgb + b) per image."""
        kernel = a_embedded[:, :9].reshape(-1, 3, 3)
        bias = a_embedded[:, 9:].reshape(-1, 3, 1)
        rgb = rgb.reshape(batch_size, -1, 3)
        rgb = torch.bmm(kernel, rgb.transpose(1, 2)) + bias
        return self.sigmoid(rgb.transpose(1, 2).reshape(-1, 3))

    def affine_color_transform(self, args, rgb, hist, batch_size):
        assert args.encode_hist and self.typ == 'coarse'
        self.a_embedded = self.exposure_coefficients(hist)
        return self.apply_affine(self.a_embedded, rgb, batch_size)


def run_network_NeRFH_NFF(inputs, viewdirs, ts, fn, embed_fn=None, embeddirs_fn=None, typ='coarse', output_transient=False,
                          netchunk=1024 * 64, test_time=False, store_rgb=False):
    """Reference signature (nerfh_nff.py:168-170).  inputs [N,S,3], viewdirs [N,3] -> raw [N,S,R].
    The embedders and `netchunk` are accepted for compatibility; the fused kernel embeds in-register and
    tiles internally.  The result is a permuted view of the kernel's channel-major raw_t [N,R,S]."""
    pk = fn.packed()
    if typ == 'coarse' and test_time:
        mode = L.FIELD_SIGMA
    elif typ == 'coarse' or not output_transient:
        mode = L.FIELD_STATIC
    else:
        mode = L.FIELD_FULL
    dev = pk.blob.device
    raw_t = ops.FieldFromPoints.apply(inputs.to(dev), None if viewdirs is None else viewdirs.to(dev), pk, mode)
    return raw_t.permute(0, 2, 1)


class _Embedder:
    """Embedder (nerfh_nff.py:234-270), kept because create_nerf returns embed fns inside the query lambda."""

    def __init__(self, num_freqs, max_freq_log2=None):
        max_freq_log2 = num_freqs - 1 if max_freq_log2 is None else max_freq_log2
        self.N_freqs, self.max_freq_log2 = num_freqs, max_freq_log2
        self.freq_bands = 2. ** torch.linspace(0., max_freq_log2, steps=num_freqs) if num_freqs > 0 else torch.zeros(0)
        self.out_dim = 3 * (1 + 2 * num_freqs)

    def embed(self, x):
        if self.max_freq_log2 == 0:             # nerfh_nff.py:264-268: "max_freq_log2 != 0 ? cat(...) : inputs" -- also for multires = 1
            return x
        parts = [x]
        for f in self.freq_bands:
            parts += [torch.sin(x * f), torch.cos(x * f)]
        return torch.cat(parts, -1)


def get_embedder(multires, i=0, reduce_mode=-1, epochToMaxFreq=-1):
    """nerfh_nff.py:303-354.  The kernels compute the paper-default embedding; modes 0 (half the octaves) and 1 (no embedding) and
    smaller `multires` are prefixes of it and run on the same kernels (NeRFH_NFF._supported).  Mode 2 (the Nerfies-style window) does
    not run in the reference's own NFF path -- run_network_NeRFH_NFF calls `embed_fn(x)` without the epoch the mode-2 lambda requires
    (nerfh_nff.py:197,211,226 vs :351) -- and is not built."""
    if i == -1:
        return nn.Identity(), 3
    if reduce_mode == 2:
        raise NotImplementedError("nefes_amd: reduce_embedding=2 (DNeRF window) is not built; the reference's NeRFH_NFF path cannot "
                                  "run it either (embed_fn is called without the epoch argument, nerfh_nff.py:197)")
    if reduce_mode == 0:
        eo = _Embedder(multires // 2, (multires - 1) // 2)
    elif reduce_mode == 1:
        eo = _Embedder(0, 0)
    else:
        eo = _Embedder(multires)
    if eo.N_freqs > 0 and not torch.equal(eo.freq_bands, 2. ** torch.arange(eo.N_freqs, dtype=eo.freq_bands.dtype)):
        raise NotImplementedError(f"nefes_amd: {eo.N_freqs} frequencies up to 2^{eo.max_freq_log2} (multires = {multires}, mode {reduce_mode}) are not the octaves 1, 2, 4, ...: "
                                  f"not a prefix of the embedding the kernels compute")
    return (lambda x, eo=eo: eo.embed(x)), eo.out_dim, eo
