"""Import shim: `import nvsr_amd` loads the package that lives in ./neural-volume-super-resolution_amd/
(the directory name required by the repo layout is not a valid Python identifier)."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "neural-volume-super-resolution_amd")
_spec = importlib.util.spec_from_file_location("nvsr_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["nvsr_amd"] = _mod
_spec.loader.exec_module(_mod)
