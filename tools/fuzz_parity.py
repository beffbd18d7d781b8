#!/usr/bin/env python3
"""Randomised parity sweep (development tool): random shapes, overlap modes, cloud / aerosol / surface options, McICA on and off with
either random number generator -- the device against the live reference Fortran (oracle/_ref) on the GPU box, or, with --emu, the
host emulation of the device functions (tests/emu; small grids) against it on any machine.
usage: tools/fuzz_parity.py [--emu] [n=60] [seed=0]"""
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    from climt_amd._lib import Context
    from climt_amd.synthetic import make_columns, overcast
    from helpers import CONSTANTS, CPDAIR, live_oracle, maxdiff
    argv = [a for a in sys.argv[1:] if a != "--emu"]
    emu = "--emu" in sys.argv[1:]
    n = int(argv[0]) if len(argv) > 0 else 60
    rng = np.random.default_rng(int(argv[1]) if len(argv) > 1 else 0)
    if emu:
        from helpers import EmuContext
        ctx = EmuContext()
    else:
        ctx = Context(0); ctx.set_constants(**CONSTANTS); ctx.sw_init(CPDAIR); ctx.lw_init(CPDAIR)
    worst = {"sw": 0.0, "lw": 0.0}
    for it in range(n):
        ncol, nlay = int(rng.choice([1, 5, 19, 37] if emu else [1, 37, 64, 129, 300, 777])), int(rng.choice([4, 11, 30, 47, 60, 75, 100]))
        mcica = bool(rng.integers(0, 2))
        c = make_columns(ncol, nlay, cloudy=True, seed=int(rng.integers(1, 10 ** 6))); c.pop("lat")
        if nlay < 8:
            c["cldfr"][1:3] = 0.5; c["cliqwp"][1:3] = 30.0; c["cicewp"][1:3] = 0.0
        if not mcica:
            c = overcast(c)
        c.update(icld=int(rng.integers(0, 4)), iaer=0, adjes=1.0, dyofyr=int(rng.integers(0, 366)), scon=float(rng.choice([0.0, 1361.0])), isolvar=0,
                 inflg=2, iceflg=int(rng.integers(1, 4)), liqflg=1, irng=int(rng.integers(0, 2)), permuteseed=int(rng.integers(1, 1024)), idrv=int(rng.integers(0, 2)))
        c["coszen"] = np.clip(c["coszen"] * rng.uniform(-0.2, 1.2, ncol), -0.1, 1.0)      # night columns too
        c["emis"] = rng.uniform(0.85, 1.0, (16, ncol))
        gases = int(rng.integers(0, 3))
        if gases == 2:    # trace gases far from the reference profiles: taumol's "too abundant" column adjustments (lw_adjcol's pow branch)
            c["co2"] = c["co2"] * rng.uniform(0.3, 9.0, ncol)[None, :]; c["n2o"] = c["n2o"] * rng.uniform(0.3, 6.0, ncol)[None, :]
            c["ch4"] = c["ch4"] * rng.uniform(0.3, 4.0, ncol)[None, :]; c["o3"] = c["o3"] * rng.uniform(0.3, 3.0, (nlay, ncol))
        opt = rng.integers(0, 4)
        if opt == 1:      # user aerosols
            c["tauaer"] = rng.uniform(0, 0.05, (14, nlay, ncol)); c["ssaaer"] = rng.uniform(0.7, 1.0, (14, nlay, ncol)); c["asmaer"] = rng.uniform(0.2, 0.8, (14, nlay, ncol))
            c["iaer"] = 10
        elif opt == 2:    # ECMWF aerosols
            c["ecaer"] = rng.uniform(0, 0.05, (6, nlay, ncol)); c["iaer"] = 6
        elif opt == 3:    # direct cloud optics
            cld = c["cldfr"] > 0
            g = rng.uniform(0.7, 0.9, (nlay, ncol, 14))
            c.update(taucld=rng.uniform(0.1, 5.0, (nlay, ncol, 14)) * cld[:, :, None], ssacld=rng.uniform(0.9, 0.99999, (nlay, ncol, 14)), asmcld=g, fsfcld=g * g, inflg=0)
        sw_in = dict(c)
        lw_in = {k: v for k, v in c.items() if k not in ("tauaer", "ssaaer", "asmaer", "ecaer", "taucld", "ssacld", "asmcld", "fsfcld")}
        lw_in["inflg"] = 2
        lw_in["tauaer"] = rng.uniform(0, 0.03, (16, nlay, ncol))
        lw_in["iceflg"] = int(rng.integers(0, 4)); lw_in["liqflg"] = int(rng.integers(0, 2))
        if lw_in["iceflg"] == 0:
            lw_in["reice"] = np.maximum(lw_in["reice"], 10.0)
        gsw, glw = ctx.sw_fluxes(sw_in, mcica=mcica), ctx.lw_fluxes(lw_in, mcica=mcica)
        # the Mersenne twister's stream runs over (sub-column, column, layer) of the WHOLE call: the reference cannot be fed in
        # column chunks then (kissvec seeds per column, clear sky draws nothing)
        chunk = ncol if (mcica and c["irng"] == 1) else 128
        rsw, _, kind = live_oracle(sw_in, mcica, chunk=chunk, procs=8, spectra=("sw",), timeout=300)
        _, rlw, _ = live_oracle(lw_in, mcica, chunk=chunk, procs=8, spectra=("lw",), timeout=300)
        dsw = max(maxdiff(gsw[k], rsw[k]) for k in rsw); dlw = max(maxdiff(glw[k], rlw[k]) for k in rlw)
        worst["sw"], worst["lw"] = max(worst["sw"], dsw), max(worst["lw"], dlw)
        flag = "" if dsw < 1e-6 and dlw < 1e-7 else "   <<<<<<"
        print("%3d %s ncol %4d nlay %3d mcica %d rng %d icld %d opt %d gases %d ice %d/%d liq %d idrv %d  |d| sw %.2e lw %.2e%s" % (
            it, kind, ncol, nlay, mcica, c["irng"], c["icld"], opt, gases, c["iceflg"], lw_in["iceflg"], lw_in["liqflg"], c["idrv"], dsw, dlw, flag), flush=True)
    print("worst:", worst)


if __name__ == "__main__":
    main()
