"""SlabSurface -- drop-in for climt.SlabSurface (climt/_components/slab_surface.py:12-438), the surface energy balance
of a slab of possibly varying heat capacity: the consumer of the surface fluxes the radiation path produces.  Same
property dictionaries and outputs; the per-column kernel (:440-517) runs on the GPU through rrtmg_hip_slab_surface
(include/rrtmg_hip.h).  The optional Ekman heat-transport terms (`include_ekman=True`, :302-395: horizontal curl /
divergence stencils on the lat-lon grid) are outside the radiation path and are not built."""
import numpy as np

from ._sympl_compat import TendencyComponent
from .rrtmg.common import make_context

# land: 0, land_ice: 1, sea: 2, sea_ice: 3 (slab_surface.py:7-9)
AREA_MAP = {"land": 0, "land_ice": 1, "sea": 2, "sea_ice": 3}


def _col(units):
    return {"dims": ["*"], "units": units}


class SlabSurface(TendencyComponent):
    """Calculate the surface energy balance of a slab surface, on AMD MI355X."""

    input_properties = {
        "downwelling_longwave_flux_in_air": {"dims": ["*", "interface_levels"], "units": "W m^-2"},
        "downwelling_shortwave_flux_in_air": {"dims": ["*", "interface_levels"], "units": "W m^-2"},
        "upwelling_longwave_flux_in_air": {"dims": ["*", "interface_levels"], "units": "W m^-2"},
        "upwelling_shortwave_flux_in_air": {"dims": ["*", "interface_levels"], "units": "W m^-2"},
        "surface_upward_latent_heat_flux": _col("W m^-2"),
        "surface_temperature": _col("degK"),
        "surface_upward_sensible_heat_flux": _col("W m^-2"),
        "surface_thermal_capacity": _col("J kg^-1 degK^-1"),
        "surface_material_density": _col("kg m^-3"),
        "upward_heat_flux_at_ground_level_in_soil": _col("W m^-2"),
        "heat_flux_into_sea_water_due_to_sea_ice": _col("W m^-2"),
        "area_type": _col("dimensionless"),
        "soil_layer_thickness": _col("m"),
        "ocean_mixed_layer_thickness": _col("m"),
        "heat_capacity_of_soil": _col("J kg^-1 degK^-1"),
        "sea_water_density": _col("kg m^-3"),
        "ocean_heat_transport_convergence": _col("W m^-2"),
    }

    tendency_properties = {"surface_temperature": _col("degK s^-1")}

    diagnostic_properties = {
        "depth_of_slab_surface": _col("m"),
        "ocean_heat_transport_convergence": _col("W m^-2"),
    }

    def __init__(self, include_ekman=False, equatorial_ekman_cap_latitude=5.0, device=0, context=None, **kwargs):
        if include_ekman:
            raise NotImplementedError("SlabSurface(include_ekman=True): the Ekman heat-transport stencils are not part of this build")
        self._include_ekman = include_ekman
        self._eq_cap = equatorial_ekman_cap_latitude
        super(SlabSurface, self).__init__(**kwargs)
        self._ctx = context if context is not None else make_context(device)

    def __call__(self, state, *args, **kwargs):
        """A host state goes through sympl's machinery to array_call; a climt_amd.DeviceState (state resident in HBM) takes
        the device path: same quantities, DeviceQuantity handles instead of arrays (climt_amd/device_state.py)."""
        from .device_state import DeviceState, slab_device_call
        if isinstance(state, DeviceState):
            return slab_device_call(self, state)
        return super(SlabSurface, self).__call__(state, *args, **kwargs)

    def array_call(self, state):
        area_type_raw = state["area_type"]
        area_type_str = np.asarray(area_type_raw).astype(str)
        area_type_code = np.zeros(area_type_str.shape, dtype=np.int32)
        for k, v in AREA_MAP.items():
            area_type_code[area_type_str == k] = v

        def flat(x):
            return np.reshape(np.asarray(x, dtype=np.float64), (-1,))

        def surface(x):          # dims ["*", "interface_levels"]: level 0 is the surface
            x = np.asarray(x, dtype=np.float64)
            return flat(x[..., 0] if x.ndim > 1 else x)

        ocean_heat_transport = flat(state["ocean_heat_transport_convergence"])
        tend_ts, depth = self._ctx.slab_surface(
            flat(area_type_code).astype(np.int32),
            sw_down=surface(state["downwelling_shortwave_flux_in_air"]), lw_down=surface(state["downwelling_longwave_flux_in_air"]),
            sw_up=surface(state["upwelling_shortwave_flux_in_air"]), lw_up=surface(state["upwelling_longwave_flux_in_air"]),
            lh=flat(state["surface_upward_latent_heat_flux"]), sh=flat(state["surface_upward_sensible_heat_flux"]),
            up_heat_soil=flat(state["upward_heat_flux_at_ground_level_in_soil"]),
            heat_flux_sea_ice=flat(state["heat_flux_into_sea_water_due_to_sea_ice"]),
            sea_water_dens=flat(state["sea_water_density"]), surf_dens=flat(state["surface_material_density"]),
            heat_cap_soil=flat(state["heat_capacity_of_soil"]), surf_therm_cap=flat(state["surface_thermal_capacity"]),
            ocean_mix_thick=flat(state["ocean_mixed_layer_thickness"]), soil_layer_thick=flat(state["soil_layer_thickness"]),
            ocean_heat_transport=ocean_heat_transport)
        shape = np.shape(area_type_raw)
        diagnostics = {
            "depth_of_slab_surface": np.reshape(depth, shape),
            "ocean_heat_transport_convergence": np.reshape(ocean_heat_transport, shape),
        }
        return {"surface_temperature": np.reshape(tend_ts, shape)}, diagnostics
