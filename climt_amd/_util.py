"""Host helpers of the RRTMG components (climt/_core/util.py:7-15, :47-86, :89-142 in the reference)."""
import functools

import numpy as np


def ensure_contiguous_state(func):
    """Make every ndarray of the raw state C-contiguous before array_call (util.py:7-15)."""
    @functools.wraps(func)
    def wrapper(self, state, *args, **kwargs):
        for name, value in state.items():
            if isinstance(value, np.ndarray):
                state[name] = np.ascontiguousarray(value)
        return func(self, state, *args, **kwargs)
    return wrapper


def mass_to_volume_mixing_ratio(mass_mixing_ratio, molecular_weight=None, molecular_weight_air=28.964):
    """g/g -> mole/mole (util.py:47-86); RRTMG passes 18.02 for water vapour."""
    if molecular_weight is None:
        raise ValueError("The molecular weight must be provided")
    return mass_mixing_ratio * molecular_weight_air / molecular_weight


def get_interface_values(mid_level_values, surface_value, mid_level_pressure, interface_level_pressure):
    """log-pressure weighted interpolation of a mid-level quantity to the interfaces; the surface takes
    `surface_value`, the top takes the top mid-level value (util.py:89-142)."""
    nlev, ncol = mid_level_values.shape
    out = np.zeros((nlev + 1, ncol), dtype=np.double)
    logp = np.log(mid_level_pressure)
    weight = (np.log(interface_level_pressure[1:-1, :]) - logp[1:, :]) / (logp[:-1, :] - logp[1:, :])
    out[1:-1, :] = mid_level_values[1:, :] - weight * (mid_level_values[1:, :] - mid_level_values[:-1, :])
    out[0, :] = surface_value[:]
    out[-1, :] = mid_level_values[-1, :]
    return out
