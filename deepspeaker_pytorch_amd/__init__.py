"""Importable alias of the package directory `deepspeaker-pytorch_amd/` (a hyphen is not a valid
Python identifier): `import deepspeaker_pytorch_amd` resolves submodules from that directory."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                          "deepspeaker-pytorch_amd")]
with open(_os.path.join(__path__[0], "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(__path__[0], "__init__.py"), "exec"))
del _os, _f
