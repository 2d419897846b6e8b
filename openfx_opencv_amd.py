"""Import alias: the package directory is named `openfx-opencv_amd` (not a valid Python identifier),
so `import openfx_opencv_amd` loads it from there."""
import importlib.util
import os
import sys

_d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "openfx-opencv_amd")
_spec = importlib.util.spec_from_file_location(__name__, os.path.join(_d, "__init__.py"), submodule_search_locations=[_d])
_m = importlib.util.module_from_spec(_spec)
sys.modules[__name__] = _m
_spec.loader.exec_module(_m)
