"""sha256 of every output array of a seeded SW + LW call (clear sky and McICA), for comparing library variants bit for bit (development tool, GPU):
  for L in a.so b.so; do RRTMG_HIP_LIB=$L python tools/micro/lib_bits.py; done   -> equal lines = equal bits"""
import hashlib, os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
os.environ.setdefault("RRTMG_HIP_ALLOW_SYNTHETIC_LW", "1")
from climt_amd._lib import Context
from climt_amd.synthetic import make_columns
K = dict(pi=np.pi, grav=9.80665, planck=6.62607004e-27, boltz=1.38064852e-16, clight=2.99792458e10, avogad=6.022140857e23, alosmt=2.6867774e19, gascon=8.3144598e7, sbcnst=5.670367e-12, secdy=86400.0)
ctx = Context(0); ctx.set_constants(**K); ctx.sw_init(1004.64); ctx.lw_init(1004.64)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
for L in (60, 33):
    for cloudy in (False, True):
        c = make_columns(N, L, cloudy=cloudy, seed=11); c.pop("lat")
        c.update(icld=1, iaer=0, dyofyr=1, scon=1367.0, isolvar=0, inflg=2, iceflg=1, liqflg=1, irng=0, permuteseed=684)
        sw = ctx.sw_fluxes(c, mcica=cloudy)
        lw = ctx.lw_fluxes(c, mcica=cloudy)
        for tag, out in (("sw", sw), ("lw", lw)):
            h = hashlib.sha256()
            for k in sorted(out):
                h.update(np.ascontiguousarray(out[k]).tobytes())
            print("%s L=%d cloudy=%d %s  sum|x| %.9e" % (tag, L, cloudy, h.hexdigest()[:16], sum(float(np.abs(out[k]).sum()) for k in out)))
