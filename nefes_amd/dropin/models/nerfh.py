"""Drop-in for the three helpers the refinement scripts import from the reference's `models.nerfh`
(script/models/nerfh.py:17-19; importers: dm/DFM_APR_refine.py:13, dm/DFM_pose_refine.py:10, dm/direct_pose_model.py:9,
utils/utils.py:371).  The reference module itself drags in `models.decoder` -> kornia and the NeRFH/NeRFW model family,
none of which is on the NeFeS (nerfh_nff) render path."""
from models.nerfh_nff import img2mse, mse2psnr, to8b  # noqa: F401
