"""Build librrtmg_hip.so for gfx950 in-tree (climt_amd/_lib/) with hipcc.

Every translation unit is compiled to its own object concurrently (the two solve kernels dominate: ~1.5 min each),
then linked.  A/B variants: RRTMG_HIP_BUILD_FLAGS="-DRRTMG_..." RRTMG_HIP_BUILD_OUT=<other .so> python climt_amd/build.py --force
"""
import glob
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "_lib", "librrtmg_hip.so")
BASE = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"]


def _newest(paths):
    return max(os.path.getmtime(p) for p in paths)


def build(force=False, verbose=True):
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.cpp")))
    hdrs = glob.glob(os.path.join(CSRC, "*.h")) + [os.path.join(HERE, "..", "include", "rrtmg_hip.h")]
    out = os.environ.get("RRTMG_HIP_BUILD_OUT", OUT)
    extra = os.environ.get("RRTMG_HIP_BUILD_FLAGS", "").split()
    if not force and os.path.exists(out) and os.path.getmtime(out) >= _newest(srcs + hdrs):
        return out
    os.makedirs(os.path.dirname(out), exist_ok=True)
    objdir = os.path.join(HERE, "_lib", "obj-" + hashlib.sha1(" ".join(extra).encode()).hexdigest()[:8])
    os.makedirs(objdir, exist_ok=True)
    newest_hdr = _newest(hdrs)
    jobs, objs = [], []
    for src in srcs:
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        objs.append(obj)
        if not force and os.path.exists(obj) and os.path.getmtime(obj) >= max(os.path.getmtime(src), newest_hdr):
            continue
        cmd = BASE + extra + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        jobs.append((cmd, subprocess.Popen(cmd)))
    for cmd, p in jobs:
        if p.wait() != 0:
            raise subprocess.CalledProcessError(p.returncode, cmd)
    link = BASE + ["-shared", "-o", out] + objs
    if verbose:
        print(" ".join(link), flush=True)
    subprocess.check_call(link)
    return out


if __name__ == "__main__":
    build(force="--force" in sys.argv)
