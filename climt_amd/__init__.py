"""climt_amd -- MI355X-native RRTMG longwave + shortwave radiation, drop-in for
climt.RRTMGLongwave / climt.RRTMGShortwave (climt/_components/rrtmg/__init__.py:1-4), plus the zenith-angle producer
upstream of the shortwave, climt.Instellation and climt.BergerSolarInsolation, and the consumer of the surface
fluxes downstream, climt.SlabSurface; and what a model script needs to set the path up: get_grid / get_default_state
(climt/_core/initialization.py), UpdateFrequencyWrapper and the AdamsBashforth tendency stepper."""
from ._lib import Context, RRTMGError  # noqa: F401
from .berger import BergerSolarInsolation  # noqa: F401
from .device_state import DeviceAdamsBashforth, DeviceQuantity, DeviceState  # noqa: F401
from .initialization import get_default_state, get_grid  # noqa: F401
from .instellation import Instellation  # noqa: F401
from .rrtmg import RRTMGLongwave, RRTMGShortwave  # noqa: F401
from .slab_surface import SlabSurface  # noqa: F401
from .timestepping import AdamsBashforth  # noqa: F401
from .wrappers import UpdateFrequencyWrapper  # noqa: F401

__all__ = ["RRTMGLongwave", "RRTMGShortwave", "Instellation", "BergerSolarInsolation", "SlabSurface", "get_grid", "get_default_state", "UpdateFrequencyWrapper", "AdamsBashforth",
           "Context", "RRTMGError", "DeviceState", "DeviceQuantity", "DeviceAdamsBashforth"]
