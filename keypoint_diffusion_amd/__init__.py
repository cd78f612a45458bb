"""Import alias: the product package lives in `keypoint-diffusion_amd/` (a directory name
Python cannot import directly).  This shim makes it importable as `keypoint_diffusion_amd`."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), 'keypoint-diffusion_amd')
__path__ = [_real]
with open(_os.path.join(_real, '__init__.py')) as _f:
    exec(compile(_f.read(), _os.path.join(_real, '__init__.py'), 'exec'))
del _os, _f, _real
