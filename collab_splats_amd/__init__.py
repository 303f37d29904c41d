"""collab_splats_amd -- MI355X-native (gfx950) RaDe-GS splat rasterizer behind the
gsplat-rade call surface that BasisResearch/collab-splats uses.

Only the hot path of BASELINE.json's north_star lives here.  Compute is in
``libmisplat.so`` (hand-written HIP, C ABI: include/misplat.h); this package is the thin Python
host side.  There is no CPU fallback.
"""
from ._lib import MisplatError, load as load_library  # noqa: F401
from .rendering import rasterization  # noqa: F401
from .wrapper import fully_fused_projection, spherical_harmonics  # noqa: F401
from .strategy import DefaultStrategy, MCMCStrategy  # noqa: F401
from .ops import set_deterministic  # noqa: F401
from .optim import FusedAdam, step_all as fused_adam_step_all  # noqa: F401

__version__ = "0.1.0"


def install_gsplat_alias() -> None:
    """Make ``from gsplat.rendering import rasterization``, ``from gsplat.strategy import
    DefaultStrategy`` and ``from gsplat.cuda._wrapper import fully_fused_projection,
    spherical_harmonics`` (the reference's imports: rade_gs_model.py:15-20) resolve to this build."""
    import sys
    import types
    from . import rendering, strategy, wrapper

    if "gsplat" in sys.modules and not getattr(sys.modules["gsplat"], "__misplat_alias__", False):
        raise RuntimeError("a real 'gsplat' is already imported; refusing to shadow it")
    pkg = types.ModuleType("gsplat")
    pkg.__misplat_alias__ = True
    pkg.__path__ = []
    pkg.__version__ = "1.5.0+misplat"
    cuda = types.ModuleType("gsplat.cuda")
    cuda.__path__ = []
    cuda._wrapper = wrapper
    pkg.rendering, pkg.strategy, pkg.cuda = rendering, strategy, cuda
    pkg.rasterization = rendering.rasterization
    pkg.DefaultStrategy = strategy.DefaultStrategy
    pkg.MCMCStrategy = strategy.MCMCStrategy

    def _missing(modname):
        def __getattr__(name):
            if name.startswith("__"):
                raise AttributeError(name)
            raise ImportError(f"'{modname}.{name}' is not provided by collab_splats_amd's gsplat alias: only the "
                              f"RaDe-GS rasterizer path is built (rasterization, fully_fused_projection, "
                              f"spherical_harmonics, DefaultStrategy; MCMCStrategy is a placeholder)")
        return __getattr__

    pkg.__getattr__ = _missing("gsplat")
    cuda.__getattr__ = _missing("gsplat.cuda")
    sys.modules.update({"gsplat": pkg, "gsplat.rendering": rendering, "gsplat.strategy": strategy,
                        "gsplat.cuda": cuda, "gsplat.cuda._wrapper": wrapper})
