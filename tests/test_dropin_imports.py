"""The drop-in boundary: with `nefes_amd/dropin` ahead of the reference's `script/` directory, the
LITERAL import lines of the reference's scripts resolve -- path modules to the drop-ins, everything else to the
reference's own files.  Runs in a child interpreter (it rewires sys.path / sys.modules); skipped where the reference
tree is absent (the GPU box).  Third-party packages of the reference's environment that this image lacks
(configargparse, lietorch, kornia, ...) are replaced by empty stub modules, exactly as tools/make_goldens.py does."""
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/script"

pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present")

PRELUDE = f"""
import sys, types, importlib.machinery
ROOT, REF = {ROOT!r}, {REF!r}
sys.path[:0] = [ROOT, ROOT + "/nefes_amd/dropin", REF]
def stub(name, **attrs):
    if name in sys.modules:
        return sys.modules[name]
    try:
        return __import__(name)
    except Exception:
        pass
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    m.__path__ = []
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m
class _Any:
    def __init__(self, *a, **k): pass
    def __call__(self, *a, **k): return _Any()
    def __getattr__(self, n): return _Any()
# packages of the reference's conda environment that are not in this image (SURVEY.md Appendix C/D)
stub("configargparse", ArgumentParser=_Any)
stub("tinycudann", Network=_Any, Encoding=_Any, NetworkWithInputEncoding=_Any)
stub("lietorch", SE3=_Any, SO3=_Any)
stub("imageio"); stub("cv2"); stub("kornia"); stub("kornia.filters", filter2d=_Any)
tv = stub("torchvision"); stub("torchvision.utils", save_image=_Any, make_grid=_Any)
stub("torchvision.transforms", Compose=_Any); stub("torchvision.models"); stub("torchvision.datasets")
stub("torchvision.datasets.folder", default_loader=_Any)
stub("efficientnet_pytorch", EfficientNet=_Any); stub("torchsummary", summary=_Any); stub("transforms3d")
stub("transforms3d.quaternions"); stub("transforms3d.euler"); stub("pytorch3d"); stub("pytorch3d.transforms")
stub("skimage"); stub("skimage.metrics"); stub("matplotlib"); stub("matplotlib.pyplot"); stub("tqdm", tqdm=_Any, trange=_Any)
"""


def run(body):
    code = PRELUDE + textwrap.dedent(body)
    env = dict(os.environ, PYTHONPATH="")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=REF, env=env, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + "\n" + r.stderr[-4000:]
    return r.stdout


def test_run_nefes_import_lines():
    """script/run_nefes.py:10-13,19 verbatim."""
    out = run("""
        from models.ray_utils import get_rays_batch
        from models.options import config_parser
        from models.rendering import render, render_test, render_path, render_test_upsample, render_path_with_feature
        from models.nerfh_nff import img2mse, mse2psnr
        from models.losses import loss_dict
        import models, models.options, models.losses, models.rendering, models.nerfh_nff, models.ray_utils
        here = ROOT + "/nefes_amd/dropin/models"
        assert models.rendering.__file__.startswith(here) and models.nerfh_nff.__file__.startswith(here)
        assert models.ray_utils.__file__.startswith(here)
        assert models.options.__file__.startswith(REF) and models.losses.__file__.startswith(REF)
        import inspect
        # signatures of the reference's validation helpers (rendering.py:246,320,416,459,521)
        want = {"render_path": ["args", "render_poses", "hwf", "chunk", "render_kwargs", "gt_imgs", "savedir", "render_factor",
                                "single_gt_img", "img_ids"],
                "render_test": ["args", "train_dl", "val_dl", "hwf", "start", "render_kwargs_test", "feat_model", "pose_param_net"],
                "render_path_upsample": ["args", "render_poses", "hwf", "chunk", "render_kwargs", "gt_imgs", "savedir",
                                         "render_factor", "single_gt_img", "img_ids", "target_size"],
                "render_test_upsample": ["args", "val_dl", "hwf", "render_kwargs_test", "target_size"],
                "render_path_with_feature": ["args", "render_poses", "hwf", "chunk", "render_kwargs", "gt_imgs", "savedir",
                                             "render_factor", "single_gt_img", "img_ids", "feat_model", "global_step"],
                "render": ["H", "W", "focal", "chunk", "rays", "c2w", "ndc", "near", "far", "use_viewdirs", "c2w_staticcam",
                           "img_idx"],
                "batchify_rays": ["rays_flat", "chunk"],
                "sample_pdf": ["bins", "weights", "N_samples", "det", "pytest"]}
        for name, params in want.items():
            got = list(inspect.signature(getattr(models.rendering, name)).parameters)
            assert got[:len(params)] == params, (name, got)
        print("ok")
    """)
    assert "ok" in out


def test_refinement_import_lines():
    """dm/DFM_APR_refine.py:13-14, dm/DFM_pose_refine.py:10-12, dm/direct_pose_model.py:9-10, utils/utils.py:371."""
    out = run("""
        from models.nerfh import img2mse, mse2psnr
        from models.rendering import render
        from models.nerfh import img2mse
        from models.poses import LearnPose
        from models.ray_utils import get_rays
        from models.nerfh import to8b
        from models.nerfh_nff import create_nerf
        import models.nerfh, models.poses
        here = ROOT + "/nefes_amd/dropin/models"
        assert models.nerfh.__file__.startswith(here) and models.poses.__file__.startswith(here)
        import inspect
        assert list(inspect.signature(LearnPose.__init__).parameters)[1:] == ["num_cams", "learn_R", "learn_t", "init_c2w", "lietorch"]
        import torch
        assert abs(float(img2mse(torch.ones(4), torch.zeros(4))) - 1.0) < 1e-7
        assert abs(float(mse2psnr(torch.tensor(0.1))) - 10.0) < 1e-5
        print("ok")
    """)
    assert "ok" in out


def test_reference_modules_import_through_the_dropin():
    """The reference's own dm/* modules import with the drop-in first on the path: their `from models...` lines bind to
    the HIP-path functions."""
    out = run("""
        import dm.DFM_pose_refine as P
        import nefes_amd.render as R
        assert P.render is R.render
        from nefes_amd.pose import LearnPose
        assert P.LearnPose is LearnPose
        print("ok")
    """)
    assert "ok" in out


def test_launcher_resolves_dropin_first(tmp_path):
    """python -m nefes_amd.run_reference <script>, started from the reference's script/ directory (whose models/ would
    otherwise win): path modules resolve to the drop-ins, other models.* to the reference -- also when the reference
    directory only joins sys.path at run time (utils/set_sys_path.py appends to sys.path after start-up)."""
    script = tmp_path / "probe.py"
    script.write_text(f"import sys\nimport models.rendering as r\nsys.path.append({REF!r})\nimport models.losses as l\n"
                      "print('R', r.__file__); print('L', l.__file__); print('A', sys.argv[1:])\n")
    if not os.path.exists(os.path.join(ROOT, "nefes_amd", "libnefes_hip.so")):
        pytest.skip("library not built")
    env = dict(os.environ, PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, "-m", "nefes_amd.run_reference", str(script), "--flag", "1"], capture_output=True,
                       text=True, cwd=REF, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = dict(l.split(" ", 1) for l in r.stdout.strip().splitlines() if l[:2] in ("R ", "L ", "A "))
    assert lines["R"].startswith(os.path.join(ROOT, "nefes_amd", "dropin", "models"))
    assert lines["L"].startswith(REF)
    assert lines["A"] == "['--flag', '1']"
