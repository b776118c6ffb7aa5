"""Import shim: the product package lives in the directory `vp-suite_amd/` (the name the build contract fixes), which is
not a valid Python identifier. `import vp_suite_amd` loads that directory as the package `vp_suite_amd` — one module
object, one canonical name (pickled models refer to `vp_suite_amd.*`)."""
import importlib.util as _ilu
import os as _os
import sys as _sys

_pkg_dir = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "vp-suite_amd")
_spec = _ilu.spec_from_file_location("vp_suite_amd", _os.path.join(_pkg_dir, "__init__.py"),
                                     submodule_search_locations=[_pkg_dir])
_mod = _ilu.module_from_spec(_spec)
_sys.modules["vp_suite_amd"] = _mod
_spec.loader.exec_module(_mod)
