"""Launcher: run an UNMODIFIED script of the reference (ActiveVisionLab/NeFeS, `script/*.py`) on the HIP path.

    cd /path/to/NeFeS/script
    python -m nefes_amd.run_reference test_refinement.py --config config/7Scenes/dfnet/config_stairs_DFM.txt

`python script.py` puts the script's own directory first on sys.path, so the reference's `script/models/` would win over
any PYTHONPATH entry.  This launcher builds the search path explicitly -- [repo root, nefes_amd/dropin, script dir, ...] --
and then executes the script as `__main__`: `models.rendering`, `models.nerfh_nff`, `models.nerfh`, `models.ray_utils`
and `models.poses` resolve to the drop-in modules, every other `models.*`, `dm.*`, `utils.*`, `dataset_loaders.*`
import to the reference's own files (nefes_amd/dropin/models/__init__.py).
"""
import os
import runpy
import sys


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    if not argv:
        raise SystemExit("usage: python -m nefes_amd.run_reference <reference script.py> [script arguments...]")
    script = os.path.abspath(argv[0])
    if not os.path.isfile(script):
        raise SystemExit(f"nefes_amd.run_reference: no such script: {script}")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    dropin = os.path.join(root, "nefes_amd", "dropin")
    script_dir = os.path.dirname(script)
    head = [root, dropin, script_dir]
    sys.path[:] = head + [p for p in sys.path if os.path.abspath(p or os.getcwd()) not in {os.path.abspath(h) for h in head}]
    from nefes_amd import lib
    lib.load()                                   # fail now, loudly, if the HIP library is missing (no fallback exists)
    sys.argv = [script] + argv[1:]
    runpy.run_path(script, run_name="__main__")


if __name__ == "__main__":
    main()
