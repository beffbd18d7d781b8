import os, sys
sys.path.insert(0, os.getcwd())
import torch
torch.cuda.init()
from climt_amd import _hip
from climt_amd._lib import Context
ctx = Context(0)
from climt_amd.distributed import RcclComm
c = RcclComm(0, 1, 0)
libs = set()
for l in open("/proc/self/maps"):
    p = l.split()[-1]
    if any(k in p for k in ("amdhip64", "rccl", "rrtmg_hip", "hsa-runtime")):
        libs.add(p)
print("\n".join(sorted(libs)))
