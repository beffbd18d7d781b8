"""climt_amd -- MI355X-native RRTMG longwave + shortwave radiation, drop-in for
climt.RRTMGLongwave / climt.RRTMGShortwave (climt/_components/rrtmg/__init__.py:1-4), plus the zenith-angle producer
upstream of the shortwave, climt.Instellation and climt.BergerSolarInsolation, and the consumer of the surface
fluxes downstream, climt.SlabSurface."""
from ._lib import Context, RRTMGError  # noqa: F401
from .berger import BergerSolarInsolation  # noqa: F401
from .instellation import Instellation  # noqa: F401
from .rrtmg import RRTMGLongwave, RRTMGShortwave  # noqa: F401
from .slab_surface import SlabSurface  # noqa: F401

__all__ = ["RRTMGLongwave", "RRTMGShortwave", "Instellation", "BergerSolarInsolation", "SlabSurface", "Context", "RRTMGError"]
