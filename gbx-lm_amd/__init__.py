"""gbx_lm_amd: MI355X-native GBA QuantizedLinear hot path behind the gbx_lm API.

``from gbx_lm_amd import load, generate, generate_step, stream_generate`` mirrors
``gbx_lm/__init__.py:1-2``.  Imports are lazy so that ``import gbx_lm_amd`` (and the build entry
point) work before the HIP extension exists; calling an operator without it raises.
"""
__version__ = "0.1.0"

_API = {
    "load": "utils", "generate": "utils", "generate_step": "utils", "stream_generate": "utils",
    "load_model": "utils", "GenerationResponse": "utils",
}


def __getattr__(name):
    if name in _API:
        import importlib
        mod = importlib.import_module(f"{__name__}.{_API[name]}")
        return getattr(mod, name)
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")


__all__ = list(_API)
