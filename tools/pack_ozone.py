#!/usr/bin/env python3
"""Pack climt's 30-point reference ozone profile (climt/_data/ozone_profile.npy, a data file read by
climt/_core/initialization.py:1130-1143) into climt_amd/data/ozone_profile.npz (run in the build container only).
The pressures it is tabulated on, 1e5 * linspace(0.998, 0.001, 30) Pa, are stored beside it, ascending."""
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("CLIMT_REFERENCE", "/root/reference")


def main():
    o3 = np.load(os.path.join(REF, "climt", "_data", "ozone_profile.npy")).astype(np.float64)
    assert o3.shape == (30,)
    p = 1e5 * np.linspace(0.998, 0.001, 30)
    dst = os.path.join(ROOT, "climt_amd", "data", "ozone_profile.npz")
    np.savez(dst, pressure_Pa=p[::-1].copy(), mole_fraction=o3[::-1].copy())
    print("wrote", dst)


if __name__ == "__main__":
    main()
