from .longwave import RRTMGLongwave
from .shortwave import RRTMGShortwave

__all__ = ("RRTMGShortwave", "RRTMGLongwave")
