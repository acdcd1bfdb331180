"""MI355X-native path tracer for glTF scenes (hot path of l-johnson-code/glTF-Renderer).

The product is libmipt.so (hand-written HIP for gfx950 behind the C-ABI in include/mipt.h);
this package is the thin host-side binding plus the synthetic-scene generators used by the
tests and bench.  There is no CPU fallback: Renderer() raises if the HIP library is missing.
"""
from . import abi, camera, meshgen, scenes  # noqa: F401


def __getattr__(name):
    if name in ("Renderer", "load_library", "MiptError"):
        from . import renderer
        return getattr(renderer, name)
    raise AttributeError(name)
