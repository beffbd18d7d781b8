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


def _sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.cpp")))


def _headers():
    return sorted(glob.glob(os.path.join(CSRC, "*.h"))) + [os.path.join(HERE, "..", "include", "rrtmg_hip.h")]


def source_hash():
    """sha256 (first 16 hex digits) over the sources and headers of the library (csrc/*.hip, *.cpp, *.h and include/rrtmg_hip.h), names and contents: what the
    library was built from.  Compiled into rrtmg_hip_version() and written into every profile under profiles/ (tools/
    gpu_session.sh), so that bench.py can tell whether the committed HBM-traffic counters were measured on the library that runs."""
    h = hashlib.sha256()
    for f in _sources() + _headers():      # exactly the files build() compiles and depends on: a stray editor backup or log
        h.update(os.path.basename(f).encode() + b"\0")   # under csrc/ changes neither the library nor its hash
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def build(force=False, verbose=True):
    srcs, hdrs = _sources(), _headers()
    out = os.environ.get("RRTMG_HIP_BUILD_OUT", OUT)
    extra = os.environ.get("RRTMG_HIP_BUILD_FLAGS", "").split()
    if not force and os.path.exists(out) and os.path.getmtime(out) >= _newest(srcs + hdrs):
        return out
    src_hash = source_hash()
    os.makedirs(os.path.dirname(out), exist_ok=True)
    objdir = os.path.join(HERE, "_lib", "obj-" + hashlib.sha1(" ".join(extra).encode()).hexdigest()[:8])
    os.makedirs(objdir, exist_ok=True)
    newest_hdr = _newest(hdrs)
    jobs, objs = [], []
    for src in srcs:
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        objs.append(obj)
        # rrtmg_abi.hip carries the source hash (rrtmg_hip_version): it is compiled whenever anything is
        stale = os.path.basename(src) == "rrtmg_abi.hip"
        if not force and not stale and os.path.exists(obj) and os.path.getmtime(obj) >= max(os.path.getmtime(src), newest_hdr):
            continue
        cmd = BASE + extra + (['-DRRTMG_SRC_HASH="%s"' % src_hash] if stale else []) + ["-c", src, "-o", obj]
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
