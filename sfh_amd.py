"""Import alias for the package directory ``sports-field-homography_amd/``.

The package directory name (fixed by the repo layout contract) contains hyphens and
is therefore not a Python identifier.  ``import sfh_amd`` executes this stub, which
loads the real package from that directory under the module name ``sfh_amd`` and
replaces itself in ``sys.modules`` so that ``sfh_amd.reconstructor`` etc. resolve
to files inside ``sports-field-homography_amd/``.
"""
import importlib.util as _ilu
import os as _os
import sys as _sys

_dir = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "sports-field-homography_amd")
_spec = _ilu.spec_from_file_location(
    __name__, _os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir]
)
_mod = _ilu.module_from_spec(_spec)
_sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)
