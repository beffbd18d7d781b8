"""Instellation -- drop-in for climt.Instellation (climt/_components/instellation/component.py:9-61): zenith angle
from latitude, longitude and model time, the producer of the `zenith_angle` the shortwave consumes.  Same class name,
property dictionaries and output; the per-column kernel (component.py:85-135) runs on the GPU through
rrtmg_hip_zenith_angle (include/rrtmg_hip.h), the time arithmetic (:64-82) stays on the host."""
import datetime

import numpy as np

from ._sympl_compat import DiagnosticComponent
from .rrtmg.common import make_context


def total_days(time_diff):
    """Total time in units of days (component.py:69-76)."""
    return time_diff.days + (time_diff.seconds + time_diff.microseconds / 1000000.0) / (24 * 3600.0)


def days_from_2000(model_time):
    """Days since 2000-01-01 12:00 (component.py:64-66)."""
    return total_days(model_time - datetime.datetime(2000, 1, 1, 12, 0))


class Instellation(DiagnosticComponent):
    """Calculates the zenith angle given orbital parameters (Earth-sun system), on AMD MI355X."""

    input_properties = {
        "latitude": {"dims": ["*"], "units": "degrees_north"},
        "longitude": {"dims": ["*"], "units": "degrees_east"},
    }

    diagnostic_properties = {
        "zenith_angle": {"dims": ["*"], "units": "radians"},
    }

    def __init__(self, device=0, context=None, **kwargs):
        """`context`: share the library context (and its HIP stream) of an RRTMG component; else a new one on `device`."""
        super(Instellation, self).__init__(**kwargs)
        self._ctx = context if context is not None else make_context(device)

    def __call__(self, state, *args, **kwargs):
        """A host state goes through sympl's machinery to array_call; a climt_amd.DeviceState (state resident in HBM) takes
        the device path: same quantities, DeviceQuantity handles instead of arrays (climt_amd/device_state.py)."""
        from .device_state import DeviceState, instellation_device_call
        if isinstance(state, DeviceState):
            return instellation_device_call(self, state)
        return super(Instellation, self).__call__(state, *args, **kwargs)

    def array_call(self, state):
        lat, lon = state["latitude"], state["longitude"]
        lat_flat = np.ascontiguousarray(np.reshape(lat, (-1,)), dtype=np.float64)
        lon_flat = np.ascontiguousarray(np.reshape(lon, (-1,)), dtype=np.float64)
        julian_centuries = days_from_2000(state["time"]) / 36525.0
        zen = self._ctx.zenith_angle(lat_flat, lon_flat, julian_centuries)
        return {"zenith_angle": np.reshape(zen, np.shape(lat))}
