"""GPU sanity + timing driver (development tool): SW and LW parity vs the reference library and
device-resident timing of the LW+SW hot path on synthetic columns."""
import sys, time, numpy as np
sys.path.insert(0, '.')
from climt_amd._lib import Context, SW_OUT, LW_OUT
from climt_amd import _hip
from climt_amd.synthetic import make_columns, overcast
from oracle.ref_driver import RefSW, RefLW, CONSTANTS, CPDAIR
from tools.pack_tables import read_blob
from tools.synth_lw_tables import fill_reference_from_blob

ctx = Context(0); ctx.set_constants(**CONSTANTS); ctx.sw_init(CPDAIR); ctx.lw_init(CPDAIR)
rsw = RefSW()
blob = read_blob('climt_amd/data/rrtmg_lw_data.bin')
rlw = RefLW(); rlw.init(fill_tables=lambda r: fill_reference_from_blob(r, blob))
base = dict(icld=1, iaer=0, adjes=1.0, dyofyr=1, scon=1367.0, isolvar=0, inflg=2, iceflg=1, liqflg=1)

def cmp(name, c, mcica=False):
    r = rsw.fluxes(c, mcica=mcica); g = ctx.sw_fluxes(c, mcica=mcica)
    print('SW', name, ' '.join('%s %.1e' % (k, np.abs(g[k]-r[k]).max()) for k in g), flush=True)
    r = rlw.fluxes(c, mcica=mcica); g = ctx.lw_fluxes(c, mcica=mcica)
    print('LW', name, ' '.join('%s %.1e' % (k, np.abs(g[k]-r[k]).max()) for k in g), flush=True)

c = make_columns(200, 60); c.update(base); cmp('clear', c)
c = overcast(make_columns(200, 60, cloudy=True)); c.update(base); cmp('overcast', c)
c = make_columns(200, 60, cloudy=True); c.update(base); c.update(irng=0, permuteseed=684); cmp('mcica kiss', c, True)
c.update(irng=1, permuteseed=209652396, icld=2); cmp('mcica mt maxrand', c, True)

for N, cloudy in ((8192, False), (8192, True)):
    c = make_columns(N, 60, cloudy=cloudy); c.update(base); c.update(irng=0, permuteseed=684)
    L = 60
    dev = {k: _hip.DeviceArray.from_host(v) for k, v in c.items() if isinstance(v, np.ndarray) and k != 'lat'}
    inp = {k: v.ptr for k, v in dev.items()}; inp.update({k: v for k, v in c.items() if not isinstance(v, np.ndarray)})
    inp.update(ncol=N, nlay=L)
    so = {k: _hip.DeviceArray((L + lev, N)) for k, lev in SW_OUT}; lo = {k: _hip.DeviceArray((L + lev, N)) for k, lev in LW_OUT}
    sop = {k: v.ptr for k, v in so.items()}; lop = {k: v.ptr for k, v in lo.items()}
    for it in range(2):
        ctx.sw_fluxes(inp, mcica=cloudy, out=sop, memspace=1); ctx.lw_fluxes(inp, mcica=cloudy, out=lop, memspace=1)
    n = 5
    t = time.time()
    for it in range(n): ctx.sw_fluxes(inp, mcica=cloudy, out=sop, memspace=1)
    tsw = (time.time() - t) / n
    t = time.time()
    for it in range(n): ctx.lw_fluxes(inp, mcica=cloudy, out=lop, memspace=1)
    tlw = (time.time() - t) / n
    print('N=%d cloudy=%s: SW %.2f ms  LW %.2f ms  -> LW+SW %.0f col/s' % (N, cloudy, tsw*1e3, tlw*1e3, N/(tsw+tlw)), flush=True)
