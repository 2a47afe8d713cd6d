"""Import shim: the package directory is ``prompt-diffusion_amd/`` (the name the
build contract fixes); a hyphen is not importable, so this module forwards
``import prompt_diffusion_amd`` to it.  All sub-modules resolve through ``__path__``.
"""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                      "prompt-diffusion_amd")
__path__ = [_real]
_init = _os.path.join(_real, "__init__.py")
with open(_init) as _f:
    exec(compile(_f.read(), _init, "exec"))
del _f, _init
