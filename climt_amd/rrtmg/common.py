"""Keyword option -> RRTMG integer flag maps (same keys and values as
climt/_components/rrtmg/rrtmg_common.py:7-59) and the shared library context."""
import numpy as np

from .._lib import CONSTANT_NAMES, Context
from .._sympl_compat import get_constant

rrtmg_cloud_overlap_method_dict = {"clear_only": 0, "random": 1, "maximum_random": 2, "maximum": 3}
rrtmg_cloud_props_dict = {"direct_input": 0, "single_cloud_type": 1, "liquid_and_ice_clouds": 2}
rrtmg_cloud_ice_props_dict = {"ebert_curry_one": 0, "ebert_curry_two": 1, "key_streamer_manual": 2, "fu": 3}
rrtmg_cloud_liquid_props_dict = {"radius_independent_absorption": 0, "radius_dependent_absorption": 1}
rrtmg_aerosol_input_dict = {"no_aerosol": 0, "ecmwf": 6, "all_aerosol_properties": 10}
rrtmg_random_number_dict = {"kissvec": 0, "mersenne_twister": 1}


def physical_constants():
    """The ten constants climt hands to rrtmg[_sw]_set_constants (lw/component.py:298-309)."""
    vals = (
        np.pi,
        get_constant("gravitational_acceleration", "m/s^2"),
        get_constant("planck_constant", "erg s"),
        get_constant("boltzmann_constant", "erg K^-1"),
        get_constant("speed_of_light", "cm s^-1"),
        get_constant("avogadro_constant", "mole^-1"),
        get_constant("loschmidt_constant", "cm^-3"),
        get_constant("universal_gas_constant", "erg mol^-1 K^-1"),
        get_constant("stefan_boltzmann_constant", "W cm^-2 K^-4"),
        get_constant("seconds_per_day", "dimensionless"),
    )
    return dict(zip(CONSTANT_NAMES, vals))


_shared = {}


def make_context(device):
    """The librrtmg_hip context of `device` with the constants set -- ONE per device, shared by every component of the
    process (tables, streams and work buffers once, not once per component instance; the reference likewise keeps one set of
    module-global tables).  Flags travel with each call, so sharing is invisible to the components."""
    ctx = _shared.get(device)
    if ctx is None or not getattr(ctx, "h", None):
        ctx = _shared[device] = Context(device)
        ctx.set_constants(**physical_constants())
    return ctx
