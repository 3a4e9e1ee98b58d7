"""Drop-in `models` package: put `nefes_amd/dropin` ahead of the reference's `script/` on sys.path.

Only the modules of the render-and-refine path live here (`rendering`, `nerfh_nff`, `nerfh`, `ray_utils`, `poses`).
Every other `models.*` name the reference's scripts import (`models.options`, `models.losses`, `models.nerf`,
`models.decoder`, ... -- script/run_nefes.py:11,19) must keep resolving to the reference's own files, so the package
search path is this directory (first: its modules win) followed by every other `models/` directory found on sys.path
(the reference's `script/models/`).  The list is re-scanned on every submodule import because the reference grows
sys.path at run time (script/utils/set_sys_path.py).
"""
import os as _os
import sys as _sys

_here = _os.path.dirname(_os.path.abspath(__file__))


class _SearchPath(list):
    """`models.__path__`: [this directory] + the `models/` directories of sys.path, recomputed when iterated."""

    def _refresh(self):
        seen = {_os.path.realpath(p) for p in list.__iter__(self)}
        for entry in list(_sys.path):
            cand = _os.path.join(entry or _os.getcwd(), "models")
            real = _os.path.realpath(cand)
            if real in seen or not _os.path.isdir(cand):
                continue
            try:
                ok = any(f.endswith(".py") for f in _os.listdir(cand))
            except OSError:
                ok = False
            if ok:
                self.append(cand)
                seen.add(real)

    def __iter__(self):
        self._refresh()
        return list.__iter__(self)


__path__ = _SearchPath([_here])
