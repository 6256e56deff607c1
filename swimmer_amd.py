"""Import alias: `import swimmer_amd` loads the package that lives in the directory
`safe-exploration-with-simulator-in-rl-algorithms_amd/` (whose name is not a valid Python
identifier) as the package `swimmer_amd`."""
import importlib.util
import os
import sys

_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)),
                    "safe-exploration-with-simulator-in-rl-algorithms_amd")
_spec = importlib.util.spec_from_file_location(
    "swimmer_amd", os.path.join(_DIR, "__init__.py"), submodule_search_locations=[_DIR])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["swimmer_amd"] = _mod
_spec.loader.exec_module(_mod)
