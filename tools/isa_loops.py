#!/usr/bin/env python3
"""Static instruction mix of the loops of one kernel in a gfx950 assembly listing (development tool).

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only climt_amd/csrc/rrtmg_sw.hip -o sw.s
    python tools/isa_loops.py sw.s sw_solve_all_kernelILb0 [min_instructions]

A loop = the span between a label and a later branch back to it.  Prints, per loop, the instruction count by class so
that non-FP64 overhead (lane moves of spilled SGPRs, index arithmetic, converts, scratch traffic) stands out."""
import collections
import re
import sys


def classify(op):
    if op.startswith(("v_fma_f64", "v_mul_f64", "v_add_f64", "v_fmac_f64", "v_max_f64", "v_min_f64")): return "fp64"
    if op.startswith(("v_rcp_f64", "v_rsq_f64", "v_sqrt_f64", "v_div_", "v_ldexp_f64", "v_frexp", "v_trunc_f64", "v_floor_f64", "v_fract_f64", "v_rndne_f64")): return "fp64x"
    if op.startswith("v_cmp") or op.startswith("v_cmpx"): return "vcmp"
    if op.startswith("v_cndmask"): return "vsel"
    if op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")): return "lane"
    if op.startswith("v_cvt"): return "vcvt"
    if op.startswith(("v_mov", "v_accvgpr")): return "vmov"
    if op.startswith("v_"): return "vint"
    if op.startswith("s_waitcnt") or op.startswith("s_nop"): return "wait"
    if op.startswith(("s_cbranch", "s_branch")): return "branch"
    if op.startswith(("s_load", "s_buffer_load")): return "smem"
    if op.startswith("s_"): return "salu"
    if op.startswith("ds_"): return "lds"
    if op.startswith("scratch_"): return "scratch"
    if op.startswith(("global_load", "flat_load", "buffer_load")): return "vload"
    if op.startswith(("global_store", "flat_store", "buffer_store")): return "vstore"
    if op.startswith(("global_atomic", "flat_atomic")): return "atomic"
    return "other"


def main():
    path, kern = sys.argv[1], sys.argv[2]
    minn = int(sys.argv[3]) if len(sys.argv) > 3 else 150
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*%s\w*:" % re.escape(kern), l))
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith(".Lfunc_end"))
    body = lines[start:end]
    labels, insts = {}, []
    for l in body:
        s = l.split(";")[0].strip()
        if not s:
            continue
        m = re.match(r"^(\.LBB\d+_\d+):", s)
        if m:
            labels[m.group(1)] = len(insts)
            continue
        if s.startswith(".") or s.endswith(":"):
            continue
        insts.append(s)
    print("kernel %s: %d instructions" % (kern, len(insts)))
    tot = collections.Counter(classify(i.split()[0]) for i in insts)
    print("  whole kernel:", dict(tot.most_common()))
    loops = []
    for idx, s in enumerate(insts):
        p = s.split()
        if p[0].startswith(("s_cbranch", "s_branch")) and p[-1] in labels and labels[p[-1]] <= idx:
            loops.append((labels[p[-1]], idx, p[-1]))
    # innermost-first report of loops above the size threshold
    for a, b, lab in sorted(loops, key=lambda t: t[1] - t[0], reverse=True):
        n = b - a + 1
        if n < minn:
            continue
        c = collections.Counter(classify(i.split()[0]) for i in insts[a:b + 1])
        fp = c["fp64"] + c["fp64x"]
        print("  loop %-12s %6d instr  fp64 %5d (%4.1f%%)  %s" % (lab, n, fp, 100.0 * fp / n, dict(c.most_common())))


if __name__ == "__main__":
    main()
