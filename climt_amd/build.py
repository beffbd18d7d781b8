"""Build librrtmg_hip.so for gfx950 in-tree (climt_amd/_lib/) with hipcc."""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "_lib", "librrtmg_hip.so")


def _newest(paths):
    return max(os.path.getmtime(p) for p in paths)


def build(force=False, verbose=True):
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.cpp")))
    deps = srcs + glob.glob(os.path.join(CSRC, "*.h")) + [os.path.join(HERE, "..", "include", "rrtmg_hip.h")]
    if not force and os.path.exists(OUT) and os.path.getmtime(OUT) >= _newest(deps):
        return OUT
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    out = os.environ.get("RRTMG_HIP_BUILD_OUT", OUT)   # A/B builds: RRTMG_HIP_BUILD_FLAGS="-DRRTMG_EXACT_DIV"
    extra = os.environ.get("RRTMG_HIP_BUILD_FLAGS", "").split()
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared"] + extra + ["-o", out] + srcs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    build(force="--force" in sys.argv)
