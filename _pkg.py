"""Imports the product package (directory 'bwt-merge_amd', not a valid Python identifier)
under the module name bwt_merge_amd."""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
NAME = "bwt_merge_amd"


def load():
    if NAME in sys.modules:
        return sys.modules[NAME]
    path = os.path.join(ROOT, "bwt-merge_amd")
    spec = importlib.util.spec_from_file_location(NAME, os.path.join(path, "__init__.py"),
                                                  submodule_search_locations=[path])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[NAME] = mod
    spec.loader.exec_module(mod)
    return mod
