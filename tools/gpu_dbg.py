import sys, numpy as np
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
from climt_amd._lib import Context
from climt_amd.synthetic import make_columns
from oracle import ref_driver
from helpers import CONSTANTS, CPDAIR
BASE = dict(icld=1, iaer=0, adjes=1.0, dyofyr=1, scon=1367.0, isolvar=0, inflg=2, iceflg=1, liqflg=1)
ctx = Context(0); ctx.set_constants(**CONSTANTS); ctx.sw_init(CPDAIR)
N=int(sys.argv[1]) if len(sys.argv)>1 else 2048
c = make_columns(N, 60, cloudy=True, seed=99); c.update(BASE); c.update(irng=0, permuteseed=684)
rsw = ref_driver.RefSW()
parts=[]
for s in range(0,N,256):
    sub = {k: (v[..., s:s + 256] if isinstance(v, np.ndarray) else v) for k, v in c.items()}
    parts.append(rsw.fluxes(sub, mcica=True))
cat = {k: np.concatenate([p[k] for p in parts], axis=1) for k in ("swuflx","swdflx","swuflxc")}
g = ctx.sw_fluxes(c, mcica=True)
for k in cat:
    d=np.abs(g[k]-cat[k]).max(0); bad=np.where(d>1e-6)[0]
    print(k, 'max', d.max(), 'nbad', bad.size, bad[:10], bad[-5:] if bad.size else '')
gp=[]
for s in range(0,N,256):
    sub = {k: (v[..., s:s + 256] if isinstance(v, np.ndarray) else v) for k, v in c.items()}
    gp.append(ctx.sw_fluxes(sub, mcica=True)['swuflx'])
gc=np.concatenate(gp,axis=1)
print('gpu whole vs gpu chunks', np.abs(gc-g['swuflx']).max(), 'gpu chunks vs ref', np.abs(gc-cat['swuflx']).max())
m1=ctx.mcica_mask('sw', c['play'], c['cldfr'], 1, 684, 0)
print('mask mean', m1.mean())
