"""Drop-in without an edited line:   PYTHONPATH=/path/to/repo/shim python train.py   (or infer_*.py, train_offline.py ...)

Python imports `sitecustomize` at start-up from the first place on sys.path that has one - PYTHONPATH entries come before the
standard library - so this file runs before the reference script's first import.  It installs a meta-path finder in FRONT of
the path-based one: the first `import stable_audio_tools` / `from model import Llasa` / `from model_sigmaVAE import Llasa` /
`from flows import ...` (train.py:24, train_offline.py:19, infer_0723.py:17-18, twj_dataset.py:184-187) is answered with the
MI355X implementation (`kalle_audio_amd.install()`), although the script's own directory - sys.path[0], which holds the
reference's modules of the same names - would otherwise win.  Nothing is imported until one of those names is asked for, so
unrelated Python programs started with this PYTHONPATH are untouched.  KALLE_SHIM=0 switches it off.
"""
import importlib.abc
import importlib.util
import os
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_NAMES = ("stable_audio_tools", "model", "model_sigmaVAE", "flows")


class _Preloaded(importlib.abc.Loader):
    def __init__(self, module):
        self.module = module

    def create_module(self, spec):
        return self.module

    def exec_module(self, module):
        pass


class _KalleFinder(importlib.abc.MetaPathFinder):
    def find_spec(self, fullname, path=None, target=None):
        if fullname.split(".")[0] not in _NAMES or os.environ.get("KALLE_SHIM") == "0":
            return None
        if self in sys.meta_path:
            sys.meta_path.remove(self)                  # one shot: install() fills sys.modules for all the names at once
        if _ROOT not in sys.path:
            sys.path.append(_ROOT)                      # (appended: the package must be importable, nothing must be shadowed)
        import kalle_audio_amd
        kalle_audio_amd.install()
        mod = sys.modules.get(fullname)
        if mod is None:
            return None
        spec = importlib.util.spec_from_loader(fullname, _Preloaded(mod), is_package=hasattr(mod, "__path__"))
        return spec


if os.environ.get("KALLE_SHIM") != "0":
    sys.meta_path.insert(0, _KalleFinder())

# a sitecustomize further down sys.path (the distribution's own) keeps working: run it too
for _p in sys.path:
    _f = os.path.join(_p or ".", "sitecustomize.py")
    if os.path.isfile(_f) and os.path.abspath(_f) != os.path.abspath(__file__):
        try:
            _spec = importlib.util.spec_from_file_location("_kalle_next_sitecustomize", _f)
            _spec.loader.exec_module(importlib.util.module_from_spec(_spec))
        except Exception:       # noqa: BLE001 - exactly what site.py does with a failing sitecustomize: report, go on
            import traceback
            traceback.print_exc()
        break
