"""mrsgym_amd -- MI355X-native drop-in for the step()/reset() hot path of mrsgym ('mrs-v0').

Product package: HIP kernels behind a C-ABI (csrc/ -> lib/libmrs_hip.so) + the host-side mirror
of the reference's Gym surface.  No CPU fallback; nothing here imports oracle/.
"""
from . import native  # noqa: F401
from .native import MrsNativeError, SwarmShard, default_params, derived  # noqa: F401

__all__ = ["native", "MrsNativeError", "SwarmShard", "default_params", "derived"]
try:
    from .mrs import MRS, make  # noqa: F401
    from .reynolds import Reynolds  # noqa: F401
    from .rollout import RolloutLog  # noqa: F401
    from .analytics import MRSAnalytics  # noqa: F401
    __all__ += ["MRS", "make", "Reynolds", "RolloutLog", "MRSAnalytics"]
except ImportError:  # pragma: no cover - during bootstrap only
    pass
