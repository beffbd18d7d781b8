import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import test_gpu_parity as t
for steps in (1, 2, 3, 4):
    h = t._radiation_loop(False, steps=steps, mcica=False); d = t._radiation_loop(True, steps=steps, mcica=False)
    print("steps", steps, {n: float(np.abs(np.transpose(d[n].values, [d[n].dims.index(x) for x in h[n].dims]) - h[n].values).max()) for n in h})
