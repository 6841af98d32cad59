"""Import alias: the package lives in `omni-avsr_amd/` (the repo's mandated directory name is not a valid
Python identifier).  `import omni_avsr_amd` resolves here and forwards to that directory."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "omni-avsr_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
