"""Import shim: the package sources live in ``gbx-lm_amd/`` (not a valid Python identifier),
so ``import gbx_lm_amd`` resolves here and forwards the package path there."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "gbx-lm_amd")
if not _os.path.isdir(_real):  # pragma: no cover
    raise ImportError(f"gbx_lm_amd: package directory {_real!r} is missing")
__path__.insert(0, _real)

with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _f
