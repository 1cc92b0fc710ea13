"""Import shim: the package directory is named ``clusteredlowranksolver.jl_amd``
(a dot in a directory name is not importable with a plain ``import``), so this
module loads it from its path and registers it as ``clrs_amd``.

    import clrs_amd                      # the package
    from clrs_amd import solver, sdp     # its submodules
"""
import importlib.util
import os
import sys

_PKG_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)),
                        "clusteredlowranksolver.jl_amd")


def _load():
    name = "clrs_amd"
    spec = importlib.util.spec_from_file_location(
        name, os.path.join(_PKG_DIR, "__init__.py"),
        submodule_search_locations=[_PKG_DIR])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


_me = sys.modules.get(__name__)
if not getattr(_me, "__path__", None):
    _load()
