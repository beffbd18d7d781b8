"""climt_amd -- MI355X-native RRTMG longwave + shortwave radiation, drop-in for
climt.RRTMGLongwave / climt.RRTMGShortwave (climt/_components/rrtmg/__init__.py:1-4), plus the zenith-angle producer
upstream of the shortwave, climt.Instellation and climt.BergerSolarInsolation, and the consumer of the surface
fluxes downstream, climt.SlabSurface; and what a model script needs to set the path up: get_grid / get_default_state
(climt/_core/initialization.py), UpdateFrequencyWrapper and the AdamsBashforth tendency stepper."""
import os as _os

# The shortwave and the longwave run on two HIP streams of one context, and a communicator adds a third.  The HIP runtime
# deals streams onto GPU_MAX_HW_QUEUES hardware queues (default 4); with another library's streams in the process (torch's
# RCCL process group creates several) the two can land on ONE queue and run back to back: measured 2.00 instead of 1.76 ms
# per step.  Eight queues keep them apart.  Read by the runtime when it initialises, so it has to be in the
# environment before the first HIP call of the process; a value the user set is left alone.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

from ._lib import Context, RRTMGError  # noqa: F401,E402
from .berger import BergerSolarInsolation  # noqa: F401,E402
from .device_state import DeviceAdamsBashforth, DeviceQuantity, DeviceState  # noqa: F401,E402
from .initialization import get_default_state, get_grid  # noqa: F401,E402
from .instellation import Instellation  # noqa: F401,E402
from .rrtmg import RRTMGLongwave, RRTMGShortwave  # noqa: F401,E402
from .slab_surface import SlabSurface  # noqa: F401,E402
from .timestepping import AdamsBashforth  # noqa: F401,E402
from .wrappers import UpdateFrequencyWrapper  # noqa: F401,E402

__all__ = ["RRTMGLongwave", "RRTMGShortwave", "Instellation", "BergerSolarInsolation", "SlabSurface", "get_grid", "get_default_state", "UpdateFrequencyWrapper", "AdamsBashforth",
           "Context", "RRTMGError", "DeviceState", "DeviceQuantity", "DeviceAdamsBashforth"]
