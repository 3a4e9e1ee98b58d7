"""nefes_amd -- MI355X (gfx950) implementation of the NeFeS volumetric render-and-refine hot path.

Layout (only what the path needs):
  csrc/      hand-written HIP kernels + the C ABI (include/nefes_hip.h) -> libnefes_hip.so
  lib.py     ctypes binding of the C ABI (fails loudly when the library is missing)
  ops.py     torch.autograd.Function wrappers around the kernels
  field.py   NeRFH_NFF module mirror (reference parameter names) + packed-weight cache
  render.py  render()/render_rays()/batchify_rays() with the reference's signatures
  dist.py    ray-batch sharding across ranks + the single pose-gradient all-reduce
  dropin/    `models.{rendering,nerfh_nff,ray_utils}` modules that replace the reference's
"""
__version__ = "0.1.0"
