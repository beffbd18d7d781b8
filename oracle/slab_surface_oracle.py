"""TEST INFRASTRUCTURE ONLY -- CPU restatement (numpy) of the kernel of climt's `SlabSurface` (surface energy balance of a
slab: the consumer of the radiation path's surface fluxes).  Never imported by the product.

Follows /root/reference/climt/_components/slab_surface.py:_slab_surface_kernel_np (:440-517) -- default configuration
(include_ekman=False); area-type codes land 0, land_ice 1, sea 2, sea_ice 3 (:9).
Pinned: tests/test_oracle.py checks it against TestSlabSurface-{column,3d}-{0,1}.cache on the cached default state
(all fluxes zero there: the caches pin the masks / depths, random-input tests pin the arithmetic against this file)."""
import numpy as np

AREA_MAP = {"land": 0, "land_ice": 1, "sea": 2, "sea_ice": 3}


def slab_surface(sw_down, lw_down, sw_up, lw_up, lh, sh, area_type, up_heat_soil, heat_flux_sea_ice, sea_water_dens, surf_dens,
                 heat_cap_soil, surf_therm_cap, ocean_mix_thick, soil_layer_thick, ocean_heat_transport):
    """-> (surface temperature tendency K s^-1, depth of the slab m); all arguments 1-D over columns."""
    at = np.asarray(area_type)
    land, sea, land_ice, sea_ice = (at == 0) | (at == 1), (at == 2) | (at == 3), at == 1, at == 3
    net = sw_down + lw_down - sw_up - lw_up - sh - lh
    net = np.where(land_ice, -up_heat_soil, np.where(sea_ice, heat_flux_sea_ice, net))
    net = np.where(sea & ~sea_ice, net + ocean_heat_transport, net)
    dens = np.where(sea, sea_water_dens, surf_dens)
    depth = np.where(sea, ocean_mix_thick, np.where(land, soil_layer_thick, 0.0))
    cap = np.where(land, heat_cap_soil, surf_therm_cap)
    heat_cap_slab = (dens * depth) * cap
    with np.errstate(divide="ignore", invalid="ignore"):
        val = np.where(heat_cap_slab != 0, net / heat_cap_slab, 0.0)
    return np.where(land_ice | sea_ice, 0.0, val), depth
