"""Importable alias of the ``mm3dgs-slam_amd/`` source directory (a hyphen cannot appear in a Python module name).

``import mm3dgs_slam_amd`` executes ``mm3dgs-slam_amd/__init__.py`` with this package's ``__path__`` pointing at
that directory, so ``mm3dgs_slam_amd.rasterizer`` etc. resolve to the files stored there.
"""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "mm3dgs-slam_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _f
