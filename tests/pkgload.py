"""Import the product package: its directory name (shafa-cd_amd) has a hyphen, so it is loaded by
path under the importable alias `shafa_cd_amd`."""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load():
    if "shafa_cd_amd" in sys.modules:
        return sys.modules["shafa_cd_amd"]
    pkg_dir = os.path.join(ROOT, "shafa-cd_amd")
    spec = importlib.util.spec_from_file_location(
        "shafa_cd_amd", os.path.join(pkg_dir, "__init__.py"), submodule_search_locations=[pkg_dir])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["shafa_cd_amd"] = mod
    spec.loader.exec_module(mod)
    return mod


def load_submodule(name):
    """shafa_cd_amd.<name> (e.g. sharding) without importing torch-free parts twice."""
    import importlib
    load()
    return importlib.import_module("shafa_cd_amd." + name)
