"""Import shim: the package directory is named ``hri-emo_amd`` (not an identifier), so map the module
name ``hri_emo_amd`` onto it.  ``import hri_emo_amd`` then behaves like a normal package import."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hri-emo_amd")
_spec = importlib.util.spec_from_file_location("hri_emo_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["hri_emo_amd"] = _mod
_spec.loader.exec_module(_mod)
