"""Importable alias for the `vehicle-counting_amd/` package directory.

The package directory keeps the name the build contract asks for (`vehicle-counting_amd/`), which is
not a valid Python identifier; this stub makes `import vehicle_counting_amd` resolve to it.
"""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "vehicle-counting_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _f
