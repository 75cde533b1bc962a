"""rcognita_amd - MI355X-native implementation of rcognita's environment-step + predictive-controller
rollout path (System.closed_loop_rhs / Simulator.sim_step feeding CtrlOptPred._actor_cost /
_critic_cost), behind the reference's System / Simulator / CtrlOptPred class surface.

Compute lives in hand-written HIP kernels (rcognita_amd/csrc) behind a C ABI (include/rcg.h); this
package is the thin host side.  There is no CPU fallback: importing works anywhere the shared library
has been built, creating an engine needs a gfx950 GPU.
"""
__version__ = "0.1.0"

from . import _native  # noqa: F401
from .engine import DeviceArray, Engine, EngineConfig  # noqa: F401
