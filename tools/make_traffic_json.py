#!/usr/bin/env python3
"""profiles/hbm_traffic.json, profiles/fp64_flops.json and profiles/valu_issue.json from the PMC passes of tools/gpu_session.sh
(profiles/<round>_pmc_{clear,cloudy}_{FETCH,WRITE}_SIZE[_131072].txt, profiles/<round>_pmc_{clear,cloudy}_sq.txt).

  traffic  = (2 x FETCH_SIZE + WRITE_SIZE) KiB per launch -- FETCH_SIZE doubled per the gfx950 correction of
             MI355X_MICROARCH.md (HBM section: 128-B requests tallied at 64 B); memory-side requests of the L2s, i.e. HBM plus
             Infinity-Cache hits.  The 131072-column passes (solve chunks of RRTMG_HIP_CHUNK_TILES x 64 columns per call, prep slab 0.9-1.6 GB,
             scratch 9-14 GB per chunk: far beyond the 256 MB Infinity Cache) give the same bytes per column as the
             8192-column ones, so the traffic is HBM traffic.
  flops    = (2 x FMA_F64 + MUL_F64 + ADD_F64 + TRANS_F64) wave instructions x 64 lanes per launch.
  issue    = profiles/valu_issue.json: per solve kernel the VALU wave instructions of a launch (SQ_INSTS_VALU), the quarter-rate
             FP64 ones among them (SQ_INSTS_VALU_TRANS_F64: v_rcp / v_rsq / v_sqrt_f64) and issue_cycles = 4 x (VALU - TRANS) +
             16 x TRANS: a wave instruction occupies its SIMD's VALU for 4 cycles (16 lanes x 4 = 64; FP64 FMA / MUL / ADD run
             at that rate on gfx950: 78.6 TF = 1024 SIMDs x 16 lanes x 2 flop x 2.4 GHz), a quarter-rate one for 16.  bench.py
             divides by 1024 SIMDs and the clock: the launch's VALU issue time per SIMD, whose ratio to the kernel's duration is
             roofline.issue_frac.  (Cross-check: SQ_ACTIVE_INST_VALU of the same pass, in quad-cycles, x 4 / 1024 gives the same
             time within 2 %.)
Key: "<kernel>|<columns of the call>|<levels>|<clear|cloudy>" (what bench.py looks up); for 131072 clear-sky columns the value is
per launch of one column chunk (8192 columns), for 131072 McICA columns -- a grid with both kinds of tiles, which runs in ONE
large chunk from its second call on -- per step.
  "step|<columns>|<levels>|<clear|cloudy>" = the same counters summed over EVERY kernel of one LW+SW step (preparation,
             cloud optics / sub-column masks, both solve variants, flux + heating): sum over kernels of (average per dispatch x
             dispatches) / steps, steps = dispatches of sw_prep_fused_kernel / column chunks of a call (the preparation is
             launched once per chunk of 128 tiles = 8192 columns, the default RRTMG_HIP_CHUNK_TILES); "step_kernels|..." lists
             the terms.
Every PMC file starts with "# source_hash <hash>" (the sources the profiled library was built from); the passes of a round must
agree, and the hash goes into both files: bench.py quotes the counters only for the library they were measured on.
usage: tools/make_traffic_json.py [round=r04]"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1] if len(sys.argv) > 1 else "r04"


def get_all(fn, counter):
    """kernel -> (average per dispatch, dispatches) for every kernel in the file"""
    out = {}
    path = os.path.join(ROOT, "profiles", fn)
    if not os.path.exists(path):
        return out
    for line in open(path):
        m = re.match(r"(.*?)\s+%s\s+dispatches\s+(\d+)\s+avg\s+(\S+)" % counter, line.strip())
        if m:
            out[m.group(1).strip()] = (float(m.group(3)), int(m.group(2)))
    return out


def get(fn, counter):
    return {k: v[0] for k, v in get_all(fn, counter).items() if "_solve_" in k}


def file_hash(fn):
    try:
        first = open(os.path.join(ROOT, "profiles", fn)).readline()
    except OSError:
        return None
    return first.split()[2] if first.startswith("# source_hash") else None


hashes = set()
for fn in sorted(os.listdir(os.path.join(ROOT, "profiles"))):
    if fn.startswith(rnd + "_pmc_"):
        hashes.add(file_hash(fn))
if len(hashes) != 1 or None in hashes:
    sys.exit("the PMC passes of round %s were not all taken on one library (source hashes %s): nothing written" % (rnd, sorted(map(str, hashes))))
src_hash = hashes.pop()

traffic = {"_doc": "HBM-side bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) KiB from profiles/%s_pmc_*; see tools/make_traffic_json.py" % rnd}
issue = {"_doc": "VALU issue cycles per launch = 4 x (SQ_INSTS_VALU - SQ_INSTS_VALU_TRANS_F64) + 16 x SQ_INSTS_VALU_TRANS_F64 (wave instructions, summed over "
                 "the launch) from profiles/%s_pmc_*_sq.txt; / 1024 SIMDs / 2.4 GHz = VALU issue time per SIMD; see tools/make_traffic_json.py" % rnd}
flops = {"_doc": "FP64 flops per launch = (2 FMA + MUL + ADD + TRANS wave instructions) x 64 from profiles/%s_pmc_*_sq.txt; see tools/make_traffic_json.py" % rnd}
for mode in ("clear", "cloudy"):
    for ncol, tag in ((8192, ""), (131072, "_131072")):
        fetch, write = get("%s_pmc_%s_FETCH_SIZE%s.txt" % (rnd, mode, tag), "FETCH_SIZE"), get("%s_pmc_%s_WRITE_SIZE%s.txt" % (rnd, mode, tag), "WRITE_SIZE")
        fa, wa = get_all("%s_pmc_%s_FETCH_SIZE%s.txt" % (rnd, mode, tag), "FETCH_SIZE"), get_all("%s_pmc_%s_WRITE_SIZE%s.txt" % (rnd, mode, tag), "WRITE_SIZE")
        chunks = max(1, -(-ncol // (128 * 64)))
        steps = fa.get("rrtmg::sw_prep_fused_kernel", (0, 0))[1] // chunks
        # McICA: the sub-column mask kernel runs once per spectrum and call whatever the chunking (a grid with both kinds of
        # tiles -- the 131072-column McICA one -- switches to large chunks after its first call: launches per call vary)
        per_step = mode == "cloudy" and "rrtmg::kiss_mask_kernel" in fa
        if per_step:
            steps = fa["rrtmg::kiss_mask_kernel"][1] // 2
        for k in fetch:
            b = (2.0 * fetch[k] + write.get(k, 0.0)) * 1024.0
            if per_step and ncol > 8192 and steps:      # per STEP (= per launch of the large chunk in steady state)
                b = (2.0 * fa[k][0] * fa[k][1] + wa.get(k, (0.0, 0))[0] * wa.get(k, (0.0, 0))[1]) * 1024.0 / steps
            if b > 1.0e6:
                traffic["%s|%d|60|%s" % (k, ncol, mode)] = b
        if steps and len(fa) > 4:      # (a pass that recorded every kernel, not only the solve kernels)
            terms = {k: (2.0 * fa[k][0] * fa[k][1] + wa.get(k, (0.0, 0))[0] * wa.get(k, (0.0, 0))[1]) * 1024.0 / steps for k in fa}
            traffic["step|%d|60|%s" % (ncol, mode)] = sum(terms.values())
            traffic["step_kernels|%d|60|%s" % (ncol, mode)] = {k: v for k, v in sorted(terms.items(), key=lambda kv: -kv[1]) if v > 1.0e5}
    sq = {c: get("%s_pmc_%s_sq.txt" % (rnd, mode), c) for c in ("SQ_INSTS_VALU", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_TRANS_F64", "SQ_ACTIVE_INST_VALU")}
    for k, valu in sq["SQ_INSTS_VALU"].items():
        trans = sq["SQ_INSTS_VALU_TRANS_F64"].get(k, 0.0)
        if valu > 1.0e6:
            issue["%s|8192|60|%s" % (k, mode)] = {"valu": valu, "trans_f64": trans, "issue_cycles": 4.0 * (valu - trans) + 16.0 * trans,
                                                  "active_inst_valu_quad_cycles": sq["SQ_ACTIVE_INST_VALU"].get(k)}
    for k in sq["SQ_INSTS_VALU_FMA_F64"]:
        f = 64.0 * (2.0 * sq["SQ_INSTS_VALU_FMA_F64"][k] + sq["SQ_INSTS_VALU_MUL_F64"].get(k, 0) + sq["SQ_INSTS_VALU_ADD_F64"].get(k, 0) + sq["SQ_INSTS_VALU_TRANS_F64"].get(k, 0))
        if f > 1.0e6:
            flops["%s|8192|60|%s" % (k, mode)] = f
traffic["source_hash"] = flops["source_hash"] = issue["source_hash"] = src_hash
json.dump(traffic, open(os.path.join(ROOT, "profiles", "hbm_traffic.json"), "w"), indent=1)
json.dump(flops, open(os.path.join(ROOT, "profiles", "fp64_flops.json"), "w"), indent=1)
json.dump(issue, open(os.path.join(ROOT, "profiles", "valu_issue.json"), "w"), indent=1)
print(json.dumps(traffic, indent=1)); print(json.dumps(flops, indent=1)); print(json.dumps(issue, indent=1))
