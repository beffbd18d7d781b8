#!/usr/bin/env python3
"""bench.py -- RRTMG LW+SW columns/s on MI355X (BASELINE.json metric).

A "step" is one pass of the hot path -- shortwave + longwave fluxes and heating rates -- over one batch of synthetic
columns that is already resident in HBM.  Workload (per GPU; columns shard embarrassingly, weak scaling):

    default      BASELINE.json configs[1]: clear sky, 128x64 columns x 60 levels
    --cloudy     configs[2]: McICA liquid+ice clouds (kissvec), 128x64 x 60
    --config 4   configs[3]'s per-GPU shard: 16 384 columns x 60 levels, McICA  (8 GPUs = 512x256x60)
    --config 5   configs[4]'s per-GPU shard: 129 600 columns x 100 levels, McICA (8 GPUs = 1440x720x100)

For N>1 the output arrays are reassembled with RCCL (north_star): climt_amd.distributed.ShardedRadiation, librccl bound
through ctypes, gather of one flat double-buffered device buffer on its own stream, inside the timed region.  ONE run
measures every gather mode with the same brackets -- `all` (ncclAllGather), `direct` (the same result by one grouped
ncclSend/ncclRecv exchange, a block on each xGMI link at once), `root`, `none` -- and prints them side by side
(`gather_modes`: columns/s, the bytes a GPU receives per step, the rate that is and the rate the compute alone would need);
the headline `value` is the mode `--gather` names: a fixed mode, `all` by default (`auto` = the faster of `all` / `direct`, a
best-of-two selection that is not comparable with a fixed-mode line).  torch is used ONLY for the launch contract (process group, barrier, max over ranks); the
compute path is librrtmg_hip.so and the communicator librccl.so, both through ctypes (if librccl cannot be brought up on every
rank the run goes on without the gather and the line says so; `--comm torch` is a testing option, tests/torch_comm.py).

`python bench.py --gpus N` with N > 1 and no launcher (no WORLD_SIZE in the environment) starts the N ranks itself, one process
per GPU, exactly as the driver's `torch.distributed.run` line would; it refuses to run when the box has fewer GPUs.

Prints ONE JSON line on rank 0 (see the driver contract), with
  roofline     : dominant kernel = the solve kernel (SW or LW) that takes longer WITH THE GPU TO ITSELF (HIP events around its
                 launches on the stream it runs on, rrtmg_hip_kernel_ms; that is also the order of the kernels' shares in
                 profiles/*_kernel_stats.txt); achieved = its ALGORITHMIC bytes per launch (SW (34L+11)*8 B, LW (56L+22)*8 B
                 per column, SURVEY.md 8d, x columns per launch) / its average launch duration; a call launches a solve
                 kernel once per chunk of <= 8192 columns, durations are summed over the chunks and divided by their
                 number; peak = 8 TB/s HBM3E.  roofline.kernels lists BOTH solve kernels with their durations in the timed
                 region (SW || LW: a bracket there also holds the time workgroups waited for CUs the other stream held)
                 and alone, bytes, traffic (FETCH x 2 + WRITE from the PMC passes under profiles/) and FP64 flops;
                 step_traffic = the same counters summed over EVERY kernel of a step, step_hbm_side_frac = that / ms_per_step
                 / 6.3 TB/s (what this part sustains): how close the step is to a floor made of bytes it moves.
  cpu_baseline : the reference Fortran (oracle/_ref; LW on the synthetic k-tables) timed on the host cores of this box on a
                 bounded sample of the same columns (rank 0, N=1 only).
                 bound = "hbm": the roofline achieved / peak / frac are quoted against, as the contract defines them (hbm | mfma;
                 there is no matrix arithmetic on this path).  limiter = what actually holds the dominant kernel (VALU issue for
                 the shortwave kernels, the vector-memory pipeline of a CU for the longwave ones: docs/EXPERIMENTS.md);
                 issue_frac = its VALU issue time -- (4 x full-rate + 16 x quarter-rate FP64 wave instructions, SQ pass under
                 profiles/) / 1024 SIMDs / 2.4 GHz -- over its duration: the fraction that steers work on this path.
  mcica        : (N=1, default run) BASELINE.json configs[2] -- McICA liquid+ice clouds, 8192 x 60 -- as a CO-HEADLINE: the same
                 bracket discipline (K-step brackets repeated to --min-seconds), with its own top-level `roofline_mcica`.
  cpu_baseline : the reference Fortran (see above).
  extra        : (N=1) the end-to-end rates that include PCIe: the host-pointer C-ABI and the drop-in component classes on a
                 sympl-style state; the model step resident on the device.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CONSTANTS = dict(pi=np.pi, grav=9.80665, planck=6.62607004e-27, boltz=1.38064852e-16, clight=2.99792458e10,
                 avogad=6.022140857e23, alosmt=2.6867774e19, gascon=8.3144598e7, sbcnst=5.670367e-12, secdy=86400.0)
CPDAIR = 1004.64
HBM_PEAK = 8.0e12
HBM_SUSTAINED = 6.3e12     # what a READ-streaming kernel achieves on this part (MI355X_MICROARCH.md; tools/micro/hbm_stream: 6.5-7.2e12)
# ... and what it sustains for the step's own mix -- the scratch slabs and partial planes are written once and read once, 45 % of the
# step's HBM-side bytes are writes: copy kernels and the slab pattern itself (thousands of wavefronts each writing, then reading
# back, its own region in 1 KB runs, non-temporal) reach 4.6-5.4e12 read + written (profiles/r06_hbm_stream.txt)
HBM_SUSTAINED_MIXED = 5.3e12
# vector FP64: AMD's MI355X specification (78.6 TFLOP/s) = 256 CUs x 4 SIMDs x 16 FP64 FMA lanes x 2 flop x 2.4 GHz -- the CU count
# and clock are in MI355X_MICROARCH.md, the per-SIMD FP64 rate (a wavefront's FP64 FMA issues over 4 cycles) is AMD's CDNA figure
FP64_PEAK = 78.6e12
N_SIMD, CLOCK_HZ = 1024, 2.4e9      # 256 CUs x 4 SIMDs; peak engine clock (MI355X_MICROARCH.md): the VALU issue floor below is per SIMD
# what holds each solve kernel (measured: docs/EXPERIMENTS.md B-D), printed as roofline.limiter
LIMITER = {"sw": "valu-issue (FP64): a workgroup is alone on its CU and its 4 waves per SIMD issue VALU instructions for ~63 % of its lifetime; the kernel's time follows its VALU instruction count as long as nothing spills; its scratch slab, partial planes and prep rows cost it 0-3 % (clean ablations, round 6: docs/EXPERIMENTS.md E)",
         "lw": "latency of a layer's four dependent memory trips at 2 waves per SIMD (VALU issue 27 %; a third wave per SIMD adds nothing: round 6) + its scratch-slab STORES (1.1 GB per launch) in the same in-order vector-memory path: they cost 17 % of the kernel and 8.5 % of the step, the reads nothing; held in the L2 instead of HBM they still cost 12 % (clean ablations: profiles/r06_ab_clean_ablations.txt)"}
FLAGS = dict(icld=1, iaer=0, dyofyr=1, scon=1367.0, isolvar=0, inflg=2, iceflg=1, liqflg=1, irng=0, permuteseed=684)


def _cpu_worker(args):
    """One host process: reference Fortran SW + LW on a chunk of columns (process-global Fortran state)."""
    kind, ncol, nlay, cloudy, seed, reps = args
    import time as _t
    sys.path.insert(0, ROOT)
    from climt_amd.synthetic import make_columns
    c = make_columns(ncol, nlay, cloudy=cloudy, seed=seed)
    c.update(FLAGS)
    if kind == "reference":
        from oracle.ref_driver import RefLW, RefSW
        from tools.pack_tables import read_blob
        from tools.synth_lw_tables import fill_reference_from_blob
        sw = RefSW(); sw.init()
        blob = read_blob(os.path.join(ROOT, "climt_amd", "data", "rrtmg_lw_data.bin"))
        lw = RefLW(); lw.init(fill_tables=lambda r: fill_reference_from_blob(r, blob))
    else:
        from oracle.port_driver import PortLW, PortSW   # C restatement
        sw, lw = PortSW(), PortLW()
    sw.fluxes(c, mcica=cloudy); lw.fluxes(c, mcica=cloudy)      # (first call: page faults of the work arrays)
    t0, calls = _t.perf_counter(), 0
    while True:      # the reference keeps (ngpt, ncol, nlay) automatics on the stack: chunks of `ncol`; `reps` = seconds of work
        sw.fluxes(c, mcica=cloudy)
        lw.fluxes(c, mcica=cloudy)
        calls += 1
        if _t.perf_counter() - t0 >= reps:
            break
    return _t.perf_counter() - t0, calls


def _host_cores():
    """Cores this process may use: the affinity mask, cut to the cgroup's CPU quota when there is one."""
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = max(1, min(n, int(np.ceil(int(quota) / int(period)))))
    except Exception:
        pass
    return max(1, n)


def cpu_baseline(nlay, cloudy):
    """Reference (or port) on ALL host cores this process may use (one process per core: the Fortran is serial and keeps
    process-global state), each process working for a fixed ~10 s on chunks of the same synthetic columns; the 16-process
    figure of the earlier rounds' lines is measured beside it (`with_16_processes`)."""
    import multiprocessing as mp
    try:
        from oracle import ref_driver
        kind = "reference" if (ref_driver.available("sw") and ref_driver.available("lw")) else "port"
    except Exception:
        kind = "port"
    per = 256 if not cloudy else 96

    def run(cores, seconds):
        ctx = mp.get_context("spawn")
        with ctx.Pool(cores) as pool:
            t0 = time.perf_counter()
            got = pool.map(_cpu_worker, [(kind, per, nlay, cloudy, 1000 + i, seconds) for i in range(cores)], chunksize=1)
            wall = time.perf_counter() - t0
        # the timed regions run concurrently (every process works for `seconds`): columns done by all / the longest region
        tmax = max(t for t, _ in got)
        return sum(calls for _, calls in got) * per / tmax, tmax, wall, sum(calls for _, calls in got)
    all_cores = _host_cores()
    try:
        v, tmax, wall, calls = run(all_cores, 10.0)
        out = {"value": v, "unit": "columns/s", "cores": all_cores, "kind": kind,
               "sample": "%d processes (every core this process may use; os.cpu_count() = %d) x ~10 s of LW+SW %s calls on chunks of %d synthetic columns x %d "
                         "levels: %d calls, longest timed region %.2f s (pool wall %.1f s); LW on synthetic k-tables"
                         % (all_cores, os.cpu_count() or 0, "McICA" if cloudy else "clear-sky", per, nlay, calls, tmax, wall)}
        if all_cores > 16:
            v16, t16, w16, c16 = run(16, 8.0)
            out["with_16_processes"] = {"value": v16, "cores": 16, "sample": "16 processes x ~8 s: %d calls of %d columns, longest timed region %.2f s" % (c16, per, t16)}
        return out
    except Exception as e:  # pragma: no cover
        return {"value": None, "unit": "columns/s", "cores": 0, "kind": kind, "sample": "cpu baseline failed: %r" % (e,)}


def _profile_json(name):
    try:
        return json.load(open(os.path.join(ROOT, "profiles", name)))
    except Exception:
        return {}


class _NoComm:
    """What is left when librccl could not be brought up on every rank: no gather (the line says so), compute still measured."""
    kind, stream = "none", None

    def __init__(self, rank, world):
        self.rank, self.world = rank, world

    def wait(self):
        pass

    def close(self):
        pass


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks (this script, RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_* set as torch.distributed.run sets them), rank 0 prints the line, the exit code is the worst rank's."""
    import socket
    import subprocess
    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    # a rank that dies must not leave the others waiting in a collective until the communicator's own timeout
    rc = 0
    while any(p.poll() is None for p in procs):
        failed = [p.returncode for p in procs if p.poll() not in (None, 0)]
        if failed:
            rc = abs(failed[0])
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            break
        time.sleep(0.2)
    for p in procs:
        try:
            p.wait(timeout=30)
        except Exception:
            p.kill()
    sys.exit(rc or max(abs(p.returncode or 0) for p in procs))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default: enough for >= 1 s of timed region)")
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--config", type=int, default=2, choices=[2, 3, 4, 5], help="BASELINE.json configs[] index + 1 (2 clear, 3 McICA, 4 / 5 the 8-GPU grids' per-GPU shard)")
    ap.add_argument("--columns", type=int, default=None, help="columns per GPU (overrides the preset)")
    ap.add_argument("--levels", type=int, default=None)
    ap.add_argument("--cloudy", action="store_true", help="configs[2]: McICA liquid+ice clouds (kissvec)")
    ap.add_argument("--gather", default="all", choices=["auto", "all", "direct", "root", "none"],
                    help="N>1: which gather mode the headline value is quoted on (every mode is measured and printed either way): a FIXED "
                         "mode, `all` (ncclAllGather, what ShardedRadiation does by default) unless asked otherwise; "
                         "auto = the faster of `all` and `direct` (a best-of-two: comparable only with other auto lines)")
    ap.add_argument("--gather-modes", default="all,direct,root,none", help="N>1: the modes measured in this run (comma-separated; the headline mode is added)")
    ap.add_argument("--no-gather", action="store_true", help="same as --gather none")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the end-to-end extras (N=1)")
    ap.add_argument("--no-mcica", action="store_true", help="skip the McICA co-headline of the default run (N=1)")
    ap.add_argument("--clear-every", type=int, default=0, help="McICA workloads: clouds removed from every N-th 64-column tile -- a grid with both kinds of tiles (docs/EXPERIMENTS.md D)")
    ap.add_argument("--lw-first", action="store_true", help="enqueue the longwave before the shortwave (N=1; experiment)")
    ap.add_argument("--serial", action="store_true", help="synchronous SW then LW calls (no SW||LW stream overlap)")
    ap.add_argument("--sync-every-step", action="store_true", help="host synchronize after every step inside the timed brackets too (N=1)")
    ap.add_argument("--comm", default="rccl", choices=["rccl", "torch"], help="N>1 communicator: librccl via ctypes (default); torch = testing only (tests/torch_comm.py)")
    ap.add_argument("--dist-backend", default="nccl", help="torch.distributed backend of the launch contract (nccl = RCCL)")
    ap.add_argument("--share-device", action="store_true", help="testing: every rank uses GPU 0 (2 ranks on a 1-GPU box; with --dist-backend gloo)")
    ap.add_argument("--force-dist", action="store_true", help="testing: run the N>1 code path (communicator, gather) with a single rank too")
    ap.add_argument("--no-unpack", action="store_true", help="N>1: leave the gathered outputs in the collective's layout [rank][array][level][local column] "
                    "(default: a block-copy kernel behind the gather writes the boundary layout [array][level][column])")
    ap.add_argument("--rccl-channels", type=int, default=16, help="N>1: NCCL_MAX_NCHANNELS for the gather (each channel occupies a CU while it runs; "
                    "0 = RCCL's default; a value already in the environment wins)")
    ap.add_argument("--sort-columns", action="store_true", help="opt-in internal column order (rrtmg_hip_set_column_sort): cloud-free columns out of cloudy tiles; "
                    "a cloud-free column's shortwave then differs by ~1e-12 W m^-2 from the default's (docs/EXPERIMENTS.md E)")
    ap.add_argument("--contract-arrays", action="store_true", help="N=1: also hand over the band arrays SURVEY 8(d)'s contract bytes count and the synthetic columns "
                    "do not have -- zero `taucld` (LW 16 x L, SW 14 x L) and LW `tauaer` (16 x L) device arrays -- so that the kernels really read them")
    ap.add_argument("--no-comm-selftest", action="store_true", help="N>1: skip the communicator self-test (every gather mode once on a 1 KB pattern, checked on every rank, before the brackets)")
    ap.add_argument("--min-seconds", type=float, default=3.0, help="timed region: brackets of K steps are repeated until this much time is covered")
    a = ap.parse_args()
    preset = {2: (8192, 60, False), 3: (8192, 60, True), 4: (16384, 60, True), 5: (129600, 100, True)}[a.config]
    N = a.columns or preset[0]
    L = a.levels or preset[1]
    cloudy = a.cloudy or preset[2]
    if a.no_gather:
        a.gather = "none"

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        from climt_amd import _hip as _h
        have = _h.device_count()
        if have < a.gpus and not a.share_device:
            sys.stderr.write("bench.py: --gpus %d asked for, this box has %d GPU(s): refusing to measure fewer GPUs than asked for "
                             "(run with a --gpus the box has; --share-device puts every rank on GPU 0 for testing)\n" % (a.gpus, have))
            sys.exit(2)
        spawn_ranks(a.gpus)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = 0 if "--share-device" in sys.argv else int(os.environ.get("LOCAL_RANK", "0"))
    multi = world > 1 or a.force_dist
    if world > 1:
        a.gpus = world
    if world != max(1, a.gpus):      # (cannot happen: a launcher's WORLD_SIZE overrides --gpus above)
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE %d" % (a.gpus, world))
    dist = None
    if multi:
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")      # (--force-dist without a launcher)
        os.environ.setdefault("MASTER_PORT", "29517")
        # one node: RCCL's bootstrap sockets over loopback (the container hostname may not resolve to a routable interface),
        # dmabuf IPC for the peer mappings (the host driver supports nothing else)
        os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # torch's process group creates streams of its own before the context creates its two: with the runtime's default of
        # four hardware queues the SW and LW streams then share one and the step loses its overlap (2.00 vs 1.76 ms)
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
        # RCCL's kernels take one compute unit per channel while a gather runs, next to solve kernels that fill every CU
        # (sw_solve_all_kernel<false> is one round of one workgroup per CU): bound them; the line says how many were allowed
        if a.rccl_channels > 0:
            os.environ.setdefault("NCCL_MAX_NCHANNELS", str(a.rccl_channels))
        torch.cuda.set_device(local)
        if a.dist_backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(a.dist_backend, rank=rank, world_size=world)

    from climt_amd import _hip
    from climt_amd._lib import LW_OUT, SW_OUT, Context
    from climt_amd.synthetic import make_columns
    _hip.set_device(local)
    ctx = Context(local)
    ctx.set_constants(**CONSTANTS)
    ctx.sw_init(CPDAIR)
    ctx.lw_init(CPDAIR)
    if a.sort_columns:
        ctx.set_column_sort(True)

    def columns(ncol, nlay, cld):
        c = make_columns(ncol, nlay, cloudy=cld, seed=20260927 + rank)
        if cld and a.clear_every > 0:
            free = (np.arange(ncol) // 64) % a.clear_every == 0
            for k in ("cldfr", "cicewp", "cliqwp"):
                c[k][:, free] = 0.0
        c.update(FLAGS)
        c.pop("lat")
        return c

    def resident(c):
        dev = {k: _hip.DeviceArray.from_host(v) for k, v in c.items() if isinstance(v, np.ndarray)}
        inp = {k: v.ptr for k, v in dev.items()}
        inp.update({k: v for k, v in c.items() if not isinstance(v, np.ndarray)})
        inp.update(ncol=c["play"].shape[1], nlay=c["play"].shape[0])
        return dev, inp

    def contract_inputs(dev, inp):
        """--contract-arrays: (sw inputs, lw inputs) with the zero band arrays of the contract figure resident as well"""
        nlay, ncol = inp["nlay"], inp["ncol"]
        for name, nb in (("taucld_sw", 14), ("taucld_lw", 16), ("tauaer_lw", 16)):
            dev[name] = _hip.DeviceArray((nlay * ncol * nb,))
            dev[name].zero()
        return dict(inp, taucld=dev["taucld_sw"].ptr), dict(inp, taucld=dev["taucld_lw"].ptr, tauaer=dev["tauaer_lw"].ptr)

    def device_run(ncol, nlay, cld, steps, warmup, serial):
        """Single-GPU device-resident loop: -> (ms per step from the whole region, per-step ms list, kernel ms lists)."""
        c = columns(ncol, nlay, cld)
        dev, inp = resident(c)
        inp_sw, inp_lw = contract_inputs(dev, inp) if a.contract_arrays else (inp, inp)
        sizes = [(k, (nlay + lev) * ncol) for k, lev in SW_OUT] + [(k, (nlay + lev) * ncol) for k, lev in LW_OUT]
        flat = _hip.DeviceArray((sum(s for _, s in sizes),))
        off, so, lo = 0, {}, {}
        for i, (k, s) in enumerate(sizes):
            (so if i < 6 else lo)[k] = flat.ptr + 8 * off
            off += s
        ctx.set_deferred(not serial)
        enq = [0.0, 0.0, 0]

        def enqueue():
            t = time.perf_counter()
            if a.lw_first:
                ctx.lw_fluxes(inp_lw, mcica=cld, out=lo, memspace=1)
            ctx.sw_fluxes(inp_sw, mcica=cld, out=so, memspace=1)
            enq[0] += time.perf_counter() - t
            if not a.lw_first:
                ctx.lw_fluxes(inp_lw, mcica=cld, out=lo, memspace=1)
            enq[1] += time.perf_counter() - t
            enq[2] += 1

        def step():
            enqueue()
            ctx.synchronize()
        for _ in range(warmup):
            step()
        _hip.synchronize()
        enq[0] = enq[1] = 0.0
        enq[2] = 0
        per, ksw, klw, brackets, synced = [], [], [], [], []
        covered = 0.0
        while covered < a.min_seconds * 1.05 and len(brackets) < 1000 or not brackets:      # EXACTLY `steps` steps per bracket; brackets repeated until --min-seconds are covered
            _hip.synchronize()
            t0 = time.perf_counter()
            if serial or a.sync_every_step:
                for _ in range(steps):
                    step()
            else:
                # the host does not wait between steps (nor would a model whose state is resident on the device): each
                # stream orders step i + 1 behind step i, the status flags are sticky, one synchronize closes the bracket
                for _ in range(steps):
                    enqueue()
                ctx.synchronize()
            _hip.synchronize()
            brackets.append((time.perf_counter() - t0) * 1e3 / steps)
            covered += brackets[-1] * steps * 1e-3
        ms = float(np.median(brackets))
        # per-step latency, host enqueue time and the event-timed kernel durations: a bracket with a host synchronize after
        # every step (an enqueue behind a full queue would measure the GPU, not the host)
        _hip.synchronize()
        enq[0] = enq[1] = 0.0
        enq[2] = 0
        t0 = time.perf_counter()
        for _ in range(min(steps, 200)):
            t = time.perf_counter()
            step()
            per.append((time.perf_counter() - t) * 1e3)
            ksw.append(ctx.kernel_ms("sw", cloudy=cld))
            klw.append(ctx.kernel_ms("lw", cloudy=cld))
        synced.append((time.perf_counter() - t0) * 1e3 / max(1, len(per)))
        ctx.set_deferred(False)
        ssw, slw = [], []
        for _ in range(3):      # kernel durations without the SW||LW overlap, for reference
            ctx.sw_fluxes(inp_sw, mcica=cld, out=so, memspace=1)
            ctx.lw_fluxes(inp_lw, mcica=cld, out=lo, memspace=1)
            ssw.append(ctx.kernel_ms("sw", cloudy=cld))
            slw.append(ctx.kernel_ms("lw", cloudy=cld))
        n_all = max(1, enq[2])
        return dict(ms=ms, per=per, ksw=ksw, klw=klw, ssw=ssw, slw=slw, enq_sw=enq[0] * 1e3 / n_all, enq=enq[1] * 1e3 / n_all, c=c, brackets=brackets, synced_ms=synced[0])

    def pick_steps(ncol, nlay, cld):
        """Steps per bracket: about 1.5 s worth (estimated from the large-grid rates of DESIGN.md 5); brackets repeat to --min-seconds."""
        est = ncol * (nlay / 60.0) / (2.7e6 if cld else 5.4e6)
        return max(10, int(np.ceil(min(a.min_seconds, 1.5) * 1.1 / est)))

    steps = a.steps if a.steps is not None else pick_steps(N, L, cloudy)
    warmup = a.warmup if a.warmup is not None else max(2, min(10, steps // 20))
    comm_note, comm_kind, host_wait = "", None, False

    if not multi:
        r = device_run(N, L, cloudy, steps, warmup, a.serial)
        ms, per = r["ms"], r["per"]
    else:
        import torch
        from climt_amd.distributed import GATHER_MODES, RcclComm, ShardedRadiation
        modes = [m for m in a.gather_modes.split(",") if m in GATHER_MODES]
        if a.gather != "auto" and a.gather not in modes:
            modes.append(a.gather)
        if a.gather == "auto" and not ({"all", "direct"} & set(modes)):
            modes.append("all")
        comm = None
        if a.comm == "rccl" and any(m != "none" for m in modes):
            def bcast(payload, rk, wd):      # the unique id travels over the launch contract's process group
                box = [payload]
                dist.broadcast_object_list(box, src=0)
                return box[0]
            # communicator + one small all-gather end to end, under a watchdog: a bootstrap that never returns must cost
            # the gather (the line says so), not the run
            import threading
            box = {}

            def bring_up():
                try:
                    torch.cuda.set_device(local)      # (the current device is per thread)
                    c = RcclComm(rank, world, local, broadcast=bcast)
                    src, dst = _hip.DeviceArray((8,)), _hip.DeviceArray((8 * world,))
                    c.all_gather(src.ptr, dst.ptr, 8)
                    c.wait()
                    box["comm"] = c
                except Exception as e:
                    box["err"] = e
            th = threading.Thread(target=bring_up, daemon=True)
            th.start()
            th.join(float(os.environ.get("RRTMG_BENCH_RCCL_TIMEOUT", "240")))
            if "comm" in box:
                comm = box["comm"]
            elif "err" in box:
                comm_note = "librccl via ctypes failed (%s: %s); " % (type(box["err"]).__name__, str(box["err"])[:160])
            else:
                comm_note = "librccl via ctypes did not come up within its time limit; "
            ok = torch.tensor([1 if comm is not None else 0], device="cuda:%d" % local)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)       # all ranks or none
            if int(ok.item()) == 0:
                comm = None
        alloc = None
        if comm is None and a.comm == "torch":
            # TESTING only (two ranks on one GPU, where RCCL refuses): a communicator over torch.distributed on torch-owned
            # buffers, from the test infrastructure -- the product path is librccl through ctypes (climt_amd.distributed.RcclComm)
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            from torch_comm import TorchDeviceComm
            comm = TorchDeviceComm(dist, rank, world, "cuda:%d" % local)
            alloc = comm.alloc
        elif comm is None:
            if any(m != "none" for m in modes):
                comm_note += "no output gather in this run; "
            modes = ["none"]
            comm = _NoComm(rank, world)
        comm_kind = comm.kind
        # first-execution insurance: before the brackets every gather mode runs ONCE on a 1 KB pattern (rank id in
        # every word) and is checked on every rank; the line says per mode and rank what happened -- so that the first run on
        # more than one GPU names the collective that misbehaved instead of only "gather switched off"
        comm_selftest_result = None
        if comm.kind != "none" and not a.no_comm_selftest:
            from climt_amd.distributed import comm_selftest
            try:
                mine = comm_selftest(comm, alloc=alloc, check_untouched=comm.kind == "rccl")
            except Exception as e:      # pragma: no cover
                mine = {"all": "FAIL on rank %d: %r" % (rank, e)}
            every = [None] * world
            dist.all_gather_object(every, mine)
            comm_selftest_result = {m: ("OK" if all(r.get(m) == "OK" for r in every) else [r.get(m) for r in every if r.get(m) != "OK"])
                                    for m in ("all", "direct", "root")}
            comm_selftest_result["ranks"] = world
        cols_in = columns(N, L, cloudy)

        def fence(sr):
            sr.finish()
            dist.barrier()
            torch.cuda.synchronize()

        def bracket_ms(sr, step):
            """EXACTLY `steps` steps (barrier + sync on both sides), the slowest rank's time per step"""
            fence(sr)
            t0 = time.perf_counter()
            for _ in range(steps):      # as at N=1: no host synchronize between the steps of a bracket (unless asked for)
                step(sync=a.serial or a.sync_every_step)
            fence(sr)
            t = torch.tensor([(time.perf_counter() - t0) * 1e3 / steps], dtype=torch.float64, device="cuda:%d" % local)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t.item())

        def measure(mode):
            """One gather mode, the same discipline for each: warm-up, K-step brackets repeated until --min-seconds are covered
            (by the all-reduced bracket time, which every rank holds identically: all ranks stop after the same bracket), then
            synchronized steps for the per-step latency and the kernels' event brackets."""
            # every rank brings N columns of its own (weak scaling): the blocks are tile-aligned whenever N is a whole number of tiles
            sr = ShardedRadiation(ctx, comm, N * world, L, gather=mode, allocator=alloc, force=a.force_dist, unpack=not a.no_unpack,
                                  align=64 if N % 64 == 0 else 1)
            assert sr.ncol == N, (sr.ncol, N)
            if a.serial:
                ctx.set_deferred(False)      # (ShardedRadiation switches the deferred mode on: --serial wants synchronous SW then LW calls)
            sr.set_inputs(cols_in, already_local=True)
            hw = comm.kind != "rccl" and sr.do_gather      # (nothing to wait for before a gather that does not run)
            planned = sr.gather_ingress_bytes()            # bytes THIS rank (rank 0 prints) receives per step in this mode
            err = []

            def step(sync=True):
                try:
                    return sr.step(mcica=cloudy, host_wait=hw, sync=sync or hw)
                except Exception as e:      # keep measuring the compute; the line says that this mode's gather did not run
                    if not sr.do_gather:
                        raise
                    err.append("%s: %s" % (type(e).__name__, str(e)[:200]))
                    sr.do_gather = False
                    sr.inflight = [False] * sr.nbuf
                    return sr.step(mcica=cloudy, host_wait=hw, sync=sync or hw)
            for _ in range(warmup):
                step()
            brackets, covered = [], 0.0
            while covered < a.min_seconds * 1.05 and len(brackets) < 1000 or not brackets:
                brackets.append(bracket_ms(sr, step))
                covered += brackets[-1] * steps * 1e-3
            fence(sr)
            per, ksw, klw = [], [], []
            for _ in range(min(steps, 200)):      # per-step latency and kernel durations: synchronized steps, outside the brackets
                t = time.perf_counter()
                step()
                per.append((time.perf_counter() - t) * 1e3)
                ksw.append(ctx.kernel_ms("sw", cloudy=cloudy))
                klw.append(ctx.kernel_ms("lw", cloudy=cloudy))
            fence(sr)
            return dict(mode=mode, ms=float(np.median(brackets)), brackets=brackets, per=per, ksw=ksw, klw=klw, sr=sr, host_wait=hw,
                        gathered=bool(sr.do_gather), ingress=planned, error=err[0] if err else None, unpack=bool(sr.unpack))
        got = {}
        for m in modes:
            got[m] = measure(m)
            if m != modes[-1]:
                got[m].pop("sr").close()      # (frees the mode's buffers before the next one allocates its own)
        last = got[modes[-1]]["sr"]
        # headline: the mode --gather names; auto = the faster of the two algorithms that leave the outputs on every GPU
        if a.gather == "auto":
            cand = [m for m in ("all", "direct") if m in got and got[m]["gathered"]] or [m for m in modes if got[m]["gathered"]] or ["none"]
            head = min(cand, key=lambda m: got[m]["ms"])
        else:
            head = a.gather if a.gather in got else modes[0]
        a.gather = head
        h = got[head]
        ms, per, host_wait = h["ms"], h["per"], h["host_wait"]
        ms_none = got["none"]["ms"] if "none" in got else None
        gather_modes = {}
        for m, g in got.items():
            gather_modes[m] = {
                "value": world * N / (g["ms"] * 1e-3), "unit": "columns/s", "ms_per_step": g["ms"], "brackets": len(g["brackets"]),
                "timed_region_s": float(np.sum(g["brackets"])) * steps * 1e-3, "gather_ran": g["gathered"], "error": g["error"],
                "ingress_bytes_per_gpu_per_step": g["ingress"] if g["gathered"] else 0,
                "ingress_GBps_per_gpu_achieved": (g["ingress"] / (g["ms"] * 1e-3) / 1e9) if g["gathered"] else 0.0,
                "ingress_GBps_per_gpu_needed_at_compute_rate": (g["ingress"] / (ms_none * 1e-3) / 1e9) if (ms_none and m != "none") else None,
                "slowdown_vs_none": (g["ms"] / ms_none) if ms_none else None}
        gather_modes["_note"] = ("every mode: same run, same K-step brackets (barrier + synchronize on both sides, max over ranks), repeated to --min-seconds.  "
                                 "all = ncclAllGather of the flat output buffer, direct = one grouped ncclSend/ncclRecv exchange (a block on each xGMI link at "
                                 "once; own block not copied), root = blocks to rank 0 (its ingress is quoted), none = outputs stay on their GPU.  The gather of "
                                 "step i runs under the kernels of step i+1 (double buffer): ingress achieved = bytes a GPU receives per step / ms_per_step; "
                                 "needed = the same bytes / the `none` step time, i.e. the rate at which the gather would be free")
        gather_none = dict(gather_modes["none"], note="the brackets without the output gather: value / this value = what the gather costs") if "none" in got else None
        # the two solve kernels with the GPU to themselves (synchronous SW then LW calls on this rank's block)
        fence(last)
        ctx.set_deferred(False)
        sw_o, lw_o = last._out(0)
        ssw, slw = [], []
        for _ in range(3):
            ctx.sw_fluxes(last.inp, mcica=cloudy, out=sw_o, memspace=1)
            ctx.lw_fluxes(last.inp, mcica=cloudy, out=lw_o, memspace=1)
            ssw.append(ctx.kernel_ms("sw", cloudy=cloudy))
            slw.append(ctx.kernel_ms("lw", cloudy=cloudy))
        fence(last)
        r = dict(ksw=h["ksw"], klw=h["klw"], ssw=ssw, slw=slw, enq_sw=0.0, enq=0.0, brackets=h["brackets"], gather_none=gather_none,
                 gather_modes=gather_modes, unpack=h["unpack"], gathered=h["gathered"])
        if h["error"]:
            comm_note += "output gather FAILED and was switched off (%s); " % h["error"]
    value = world * N / (ms * 1e-3)

    res = None
    if rank == 0:
        mode = "cloudy" if cloudy else "clear"
        traffic_json, flops_json = _profile_json("hbm_traffic.json"), _profile_json("fp64_flops.json")
        # the committed counters are quoted only for the library they were measured on (hash of csrc + the header, compiled in)
        from climt_amd._lib import source_hash
        lib_hash = source_hash()
        counters_note = None
        if traffic_json.get("source_hash") != lib_hash:
            counters_note = ("profiles/hbm_traffic.json was measured on sources %s, this library is built from %s: traffic / flops not quoted "
                             "(re-run the PMC passes: tools/gpu_session.sh <round> pmc pmclarge sq; tools/make_traffic_json.py <round>)"
                             % (traffic_json.get("source_hash"), lib_hash))
            traffic_json, flops_json = {}, {}
        issue_json = _profile_json("valu_issue.json") if counters_note is None else {}

        def roofline_block(N, L, cld, r, ms, launches):
            """The `roofline` object of one configuration (columns N x levels L per GPU, McICA or clear sky) from the event-timed
            kernel durations of a run `r` and the committed counters; see the module docstring."""
            mode = "cloudy" if cld else "clear"
            kernels = []
            for which, first, name, bpc, timed, alone in (
                    ("sw", not a.lw_first, "rrtmg::sw_solve_cloudy_kernel" if cld else "rrtmg::sw_solve_all_kernel<false>", (34 * L + 11) * 8, r["ksw"], r["ssw"]),
                    ("lw", a.lw_first, "rrtmg::lw_solve_all_kernel" + ("<true, false>" if cld else "<false, false>"), (56 * L + 22) * 8, r["klw"], r["slw"])):
                t_ms, s_ms = float(np.mean(timed)), float(np.mean(alone))      # each: sum over the call's chunks
                key = "%s|%d|%d|%s" % (name, N, L, mode)
                tr, fl, iss = traffic_json.get(key), flops_json.get(key), issue_json.get(key)
                # VALU issue time of the launch on one SIMD: every wave instruction occupies its SIMD's VALU for 4 cycles, a
                # quarter-rate FP64 one (v_rcp / v_rsq / v_sqrt_f64) for 16; the work is spread over all 1024 SIMDs
                issue_ms = (iss["issue_cycles"] / N_SIMD / CLOCK_HZ * 1e3) if iss else None
                kernels.append({
                    "kernel": name, "spectrum": which, "limiter": LIMITER[which], "launches_per_step": launches, "columns_per_launch": N / launches,
                    "algorithmic_bytes_per_column": bpc, "algorithmic_bytes_per_launch": bpc * N / launches,
                    "launch_ms_timed_region": t_ms / launches, "launch_ms_alone": s_ms / launches, "enqueued_first": bool(first),
                    "achieved_GBps_alone": bpc * N / (s_ms * 1e-3) / 1e9, "frac_alone": bpc * N / (s_ms * 1e-3) / HBM_PEAK,
                    "achieved_GBps_timed_region": bpc * N / (t_ms * 1e-3) / 1e9, "frac_timed_region": bpc * N / (t_ms * 1e-3) / HBM_PEAK,
                    "traffic_per_launch": tr, "traffic_over_algorithmic": (tr / (bpc * N / launches)) if tr else None,
                    "fp64_flops_per_launch": fl, "fp64_frac_alone": (fl / (s_ms / launches * 1e-3) / FP64_PEAK) if fl else None,
                    "valu_wave_instructions_per_launch": iss["valu"] if iss else None, "quarter_rate_fp64_per_launch": iss["trans_f64"] if iss else None,
                    "valu_issue_ms_per_simd": issue_ms, "issue_frac_alone": (issue_ms / (s_ms / launches)) if issue_ms else None})
            dom = max(kernels, key=lambda k: k["launch_ms_alone"])      # the kernel that takes longer with the GPU to itself
            # Its launch duration IN THE TIMED REGION is free of queueing when its spectrum is enqueued first (it starts on CUs
            # nobody holds); otherwise the bracket would also count the wait for the other stream's workgroups: take it alone.
            src = "timed_region" if (dom["enqueued_first"] or a.serial) else "alone"
            kms, bpc = dom["launch_ms_" + src], dom["algorithmic_bytes_per_column"]
            kms_serial = dom["launch_ms_alone"]
            achieved = dom["algorithmic_bytes_per_launch"] / (kms * 1e-3) / 1e9
            traffic, flops = dom["traffic_per_launch"], dom["fp64_flops_per_launch"]
            step_traffic = traffic_json.get("step|%d|%d|%s" % (N, L, mode))      # FETCH x 2 + WRITE summed over every kernel of a step
            step_bytes = (34 * L + 11) * 8 + (56 * L + 22) * 8
            # the bytes of the arrays this run really hands over (climt_amd.synthetic.make_columns + the 12 outputs): the contract
            # figure also counts `taucld` (LW 16L, SW 14L) and LW `tauaer` (16L), which the synthetic columns do not have
            # (--contract-arrays adds them as zero device arrays), and does not count the interface temperatures, which they do
            extra_lw, extra_sw = (32 * L, 14 * L) if a.contract_arrays else (0, 0)
            shipped_lw = (17 * L + (L + 1) + (L + 1) + 1 + 16 + extra_lw + 6 * L + 4) * 8      # 12 level + 5 cloud arrays, plev, tlev, tsfc, emis(16) | outputs
            shipped_sw = (13 * L + (L + 1) + 5 + extra_sw + 6 * L + 4) * 8                      # 8 level + 5 cloud arrays, plev, 4 albedos + coszen | outputs
            shipped = shipped_lw + shipped_sw
            issue_sum = sum(k["valu_issue_ms_per_simd"] for k in kernels) if all(k["valu_issue_ms_per_simd"] for k in kernels) else None
            return {"bound": "hbm", "limiter": dom["limiter"], "kernel": dom["kernel"], "achieved": achieved, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                    "frac": achieved / (HBM_PEAK / 1e9), "traffic": traffic, "kernel_ms": kms, "duration_source": src,
                    "issue_frac": (dom["valu_issue_ms_per_simd"] / kms) if dom["valu_issue_ms_per_simd"] else None,
                    "issue_frac_serial": dom["issue_frac_alone"],
                    "launches_per_step": launches, "columns_per_launch": N / launches,
                    "algorithmic_bytes_per_column": bpc, "algorithmic_bytes_per_launch": dom["algorithmic_bytes_per_launch"],
                    "sw_solve_ms": kernels[0]["launch_ms_timed_region"], "lw_solve_ms": kernels[1]["launch_ms_timed_region"],
                    "sw_solve_ms_serial": kernels[0]["launch_ms_alone"], "lw_solve_ms_serial": kernels[1]["launch_ms_alone"],
                    "kernel_ms_serial": kms_serial, "frac_serial": dom["frac_alone"],
                    "kernels": kernels,
                    # the whole LW+SW step: its algorithmic bytes ((34L+11)*8 SW + (56L+22)*8 LW per column)
                    # over the step time, and the bytes it actually moves against what the memory system sustains
                    "step_algorithmic_bytes_per_column": step_bytes,
                    "step_frac": step_bytes * N / (ms * 1e-3) / HBM_PEAK,
                    "algorithmic_bytes_shipped_per_column": shipped,
                    "algorithmic_bytes_shipped_terms": {"lw": shipped_lw, "sw": shipped_sw, "contract_arrays_resident": bool(a.contract_arrays),
                                                        "note": "arrays this run hands over and gets back, per column: LW 12 level + 5 cloud-physics arrays, plev, tlev, tsfc, emis(16), "
                                                                "6 outputs; SW 8 level + 5 cloud-physics arrays, plev, 4 albedos, coszen, 6 outputs.  The contract figure "
                                                                "(step_algorithmic_bytes_per_column, SURVEY 8d) also counts taucld (LW 16L, SW 14L) and LW tauaer (16L), which "
                                                                "the synthetic columns do not create -- `--contract-arrays` makes them resident as zeros and read -- and leaves tlev out"},
                    "step_frac_on_shipped_bytes": shipped * N / (ms * 1e-3) / HBM_PEAK,
                    "step_issue_frac": (issue_sum / ms) if issue_sum else None,
                    "step_traffic": step_traffic,
                    "step_traffic_over_algorithmic": (step_traffic / (step_bytes * N)) if step_traffic else None,
                    "step_hbm_side_frac": (step_traffic / (ms * 1e-3) / HBM_SUSTAINED) if step_traffic else None,
                    "step_hbm_side_frac_of_mixed_rate": (step_traffic / (ms * 1e-3) / HBM_SUSTAINED_MIXED) if step_traffic else None,
                    "hbm_rates_measured": {"read_streaming": HBM_SUSTAINED, "write_read_mix_of_the_step": HBM_SUSTAINED_MIXED,
                                           "source": "profiles/r06_hbm_stream.txt (tools/micro/hbm_stream.hip): reads 6.5-7.2e12 B/s, copy and write-then-read-back slab pattern 4.6-5.4e12"},
                    "step_hbm_side_frac_of_peak": (step_traffic / (ms * 1e-3) / HBM_PEAK) if step_traffic else None,
                    "fp64_flops_per_launch": flops, "fp64_frac": (flops / (kms * 1e-3) / FP64_PEAK) if flops else None,
                    "fp64_frac_serial": (flops / (kms_serial * 1e-3) / FP64_PEAK) if flops else None,
                    "host_call_ms": {"sw": r["enq_sw"], "sw+lw": r["enq"]},
                    "library_source_hash": lib_hash, "counters_source_hash": traffic_json.get("source_hash"), "counters_note": counters_note,
                    "note": "kernel = the solve kernel that takes longer with the GPU to itself (also the larger share in profiles/*_kernel_stats.txt); "
                            "achieved = its ALGORITHMIC bytes per launch / kernel_ms (HIP events around its launches, average over the launches "
                            "of a call; duration_source says whether that is the timed region or the kernel alone); bound = the roofline peak / frac "
                            "are quoted against, as the contract defines it (HBM: this path cannot approach it on algorithmic bytes, 108 FLOP/B); "
                            "limiter = what actually holds the kernel; issue_frac = VALU issue time per SIMD / kernel_ms (kernels[].issue_frac_alone for both kernels), "
                            "step_issue_frac = the two solve kernels' VALU issue time per SIMD summed / ms_per_step: the fractions that steer work here; "
                            "step_frac is quoted on the contract's bytes, step_frac_on_shipped_bytes on the arrays this run really hands over.  traffic / step_traffic = measured HBM-side bytes (2 x FETCH_SIZE + WRITE_SIZE, PMC "
                            "passes of this command committed under profiles/); step_hbm_side_frac = step_traffic / ms_per_step / 6.3 TB/s (a read stream's rate), "
                            "step_hbm_side_frac_of_mixed_rate = the same / 5.3 TB/s (what copy kernels and the slab's own write-then-read pattern reach on this part)"}
        launches = max(1, ctx.kernel_launches("sw", cloudy=cloudy))      # column chunks per call: one launch of each solve kernel per chunk
        roofline = roofline_block(N, L, cloudy, r, ms, launches)
        par = "columns sharded x%d" % world
        if world > 1:
            par += {"all": " + RCCL all-gather of the outputs (one flat buffer, double-buffered: runs under the next step's kernels)",
                    "direct": " + RCCL grouped send/recv exchange of the outputs, every GPU to its peers directly (one flat buffer, double-buffered: runs under the next step's kernels)",
                    "root": " + RCCL gather of the outputs to rank 0 (double-buffered)", "none": ", outputs stay on their GPU"}[a.gather]
        res = {
            "metric": "LW+SW columns/sec (60 lev)", "value": value, "unit": "columns/s", "n_gpus": world, "steps": steps,
            "warmup": warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "rrtmg_lw+sw_%s_%dcol_x_%dlev_per_gpu" % ("mcica_cloudy" if cloudy else "clear_sky", N, L),
                       "baseline_config": a.config if not (a.columns or a.levels) else None,
                       "columns_per_gpu": N, "levels": L, "parallelism": par,
                       "communicator": (comm_note + comm_kind) if comm_kind else None,
                       "overlap": "none (serial calls)" if a.serial else "SW || LW on two HIP streams",
                       "column_sort": bool(a.sort_columns),
                       "timed_region_s": float(np.sum(r["brackets"])) * steps * 1e-3, "brackets": len(r["brackets"]),
                       "bracket_note": "ms_per_step = median over `brackets` timed regions of exactly `steps` steps each (max over ranks per bracket)",
                       "host_sync": ("after every step" if (a.serial or a.sync_every_step or (multi and host_wait)) else
                                     "one per bracket: the host enqueues the K steps back to back (each stream orders step i+1 behind step i) and synchronizes once; ms_per_step_median / p10_p90 below come from a separate bracket with a synchronize after every step"),
                       "ms_per_step_median": float(np.median(per)),
                       "ms_per_step_p10_p90": [float(np.percentile(per, 10)), float(np.percentile(per, 90))],
                       "lw_k_tables": "synthetic (reference LW data file missing)", "sw_k_tables": "reference"},
            "roofline": roofline,
        }
        res["cpu_baseline"] = None
        if multi:
            res["config"]["rccl_max_channels"] = os.environ.get("NCCL_MAX_NCHANNELS")
            res["config"]["gathered_layout"] = ("boundary [array][level][column] (block-copy kernel behind the gather)" if r["unpack"] else
                                                "collective [rank][array][level][local column]") if r["gathered"] else None
            res["config"]["gather_mode"] = a.gather
            res["comm_selftest"] = comm_selftest_result
            res["gather_modes"] = r.get("gather_modes")
            res["extra"] = {"gather_none": r.get("gather_none"),
                            "how_to_scale": "per-GPU work is fixed (weak scaling): `--config 4` = 512x256x60 over 8 GPUs (16384 columns each), `--config 5` = "
                                            "1440x720x100 (129600 columns x 100 levels each); `--gather auto|all|direct|root|none` (headline mode; every mode in --gather-modes is measured), `--no-unpack`, `--rccl-channels N`"}
        if world == 1 and not multi and not cloudy and a.config == 2 and not (a.columns or a.levels) and not a.no_mcica:
            # CO-HEADLINE: BASELINE.json configs[2] -- McICA liquid+ice clouds (kissvec), 128x64x60 -- in the same run with the
            # same discipline as the headline: K-step brackets (the same K) repeated until --min-seconds are covered, the
            # median bracket; its own roofline object at the top level
            m = device_run(8192, 60, True, steps, warmup, a.serial)
            m_launches = max(1, ctx.kernel_launches("sw", cloudy=True))
            res["mcica"] = {"metric": res["metric"], "value": 8192 / (m["ms"] * 1e-3), "unit": "columns/s", "ms_per_step": m["ms"], "steps": steps, "warmup": warmup,
                            "workload": "rrtmg_lw+sw_mcica_cloudy_8192col_x_60lev_per_gpu", "baseline_config": 3,
                            "timed_region_s": float(np.sum(m["brackets"])) * steps * 1e-3, "brackets": len(m["brackets"]),
                            "ms_per_step_median": float(np.median(m["per"])),
                            "ms_per_step_p10_p90": [float(np.percentile(m["per"], 10)), float(np.percentile(m["per"], 90))],
                            "sub_columns": "kissvec, permuteseed 684, random overlap (icld 1); liquid + ice clouds in the layers between 300 and 850 hPa, "
                                           "cloud fraction 0 / 0.3 / 0.6 / 1 by region (SURVEY.md 8d config 3)",
                            "ratio_to_clear_sky": m["ms"] / ms}
            res["roofline_mcica"] = roofline_block(8192, 60, True, m, m["ms"], m_launches)
        if world == 1 and not multi:
            if not a.no_extra:
                extra = {}
                # (a) configs[2] is a co-headline of the default run (see below); kept here as a pointer for older readers
                if not cloudy and res.get("mcica"):
                    extra["mcica"] = {"moved": "top-level `mcica` (same bracket discipline as the headline) and `roofline_mcica`",
                                      "value": res["mcica"]["value"], "ms_per_step": res["mcica"]["ms_per_step"]}
                # (a') BASELINE configs[3] WHOLE on this one GPU: 512 x 256 x 60 McICA, cloud-free and cloudy tiles mixed (every fourth
                # tile of the synthetic field is cloud-free at this size): compacted tile lists + large chunks
                if not cloudy and a.config == 2 and not (a.columns or a.levels):
                    try:
                        save = a.min_seconds
                        a.min_seconds = min(a.min_seconds, 1.5)
                        w = device_run(131072, 60, True, 12, 2, a.serial)
                        a.min_seconds = save
                        extra["config4_whole_grid_on_one_gpu"] = {
                            "workload": "rrtmg_lw+sw_mcica_cloudy_131072col_x_60lev (512 x 256 x 60), a quarter of the 64-column tiles cloud-free",
                            "value": 131072 / (w["ms"] * 1e-3), "unit": "columns/s", "ms_per_step": w["ms"], "steps": 12, "brackets": len(w["brackets"]),
                            "solve_launches_per_step": ctx.kernel_launches("sw", cloudy=True),
                            "note": "a grid with both kinds of tiles runs in large chunks from its second call on, every solve workgroup on consecutive "
                                    "entries of its variant's compacted tile list (2.20e6 columns/s before round 5; 8 x the config-4 shard on 8 GPUs "
                                    "gives the same numbers bit for bit: tests/test_gpu_parity.py::test_the_whole_8_gpu_grids_on_one_gpu_equal_their_eight_blocks)"}
                    except Exception as e:   # pragma: no cover
                        extra["config4_whole_grid_on_one_gpu"] = {"error": repr(e)[:200]}
                # (b) end to end including PCIe: host-pointer C-ABI (H2D of every input, D2H of the 12 outputs per call)
                try:
                    c = r["c"]
                    n_e = max(3, min(20, int(0.5 / (N * L / 60 / 1.0e6)) or 3))
                    # caller-owned, pre-allocated outputs as at the reference boundary; arrays that were never
                    # touched would add ~1 ms per MB of first-touch page faults on this box (tools/micro/host_path_timing.py)
                    ho_sw = {k: np.ones((L + lev, N)) for k, lev in SW_OUT}
                    ho_lw = {k: np.ones((L + lev, N)) for k, lev in LW_OUT}
                    ctx.sw_fluxes(c, mcica=cloudy, out=ho_sw); ctx.lw_fluxes(c, mcica=cloudy, out=ho_lw)
                    t0 = time.perf_counter()
                    for _ in range(n_e):
                        ctx.sw_fluxes(c, mcica=cloudy, out=ho_sw); ctx.lw_fluxes(c, mcica=cloudy, out=ho_lw)
                    e2e = (time.perf_counter() - t0) / n_e
                    h2d = sum(v.nbytes for v in c.values() if isinstance(v, np.ndarray))
                    extra["end_to_end_host_pointers"] = {"value": N / e2e, "unit": "columns/s", "ms_per_step": e2e * 1e3, "calls": n_e,
                                                         "bytes_in_per_step": int(h2d * 2), "bytes_out_per_step": int((12 * L + 8) * 8 * N),
                                                         "note": "rrtmg_hip_{sw,lw}_fluxes with memspace=0: pageable numpy arrays in, caller-owned pre-allocated outputs back, synchronous SW then LW"}
                except Exception as e:   # pragma: no cover
                    extra["end_to_end_host_pointers"] = {"error": repr(e)[:200]}
                # (c) the drop-in component classes on a sympl-style state (unit conversion, contiguity, numpy host prep)
                try:
                    import climt_amd
                    sw_c = climt_amd.RRTMGShortwave(mcica=cloudy, random_number_generator="kissvec") if cloudy else climt_amd.RRTMGShortwave()
                    lw_c = climt_amd.RRTMGLongwave(mcica=cloudy, random_number_generator="kissvec", allow_synthetic_tables=True) if cloudy \
                        else climt_amd.RRTMGLongwave(allow_synthetic_tables=True)
                    state = climt_amd.get_default_state([sw_c, lw_c], grid_state=climt_amd.get_grid(nx=128, ny=N // 128, nz=L))
                    sw_c(state); lw_c(state)
                    n_c = 5
                    t0 = time.perf_counter()
                    for _ in range(n_c):
                        sw_c(state); lw_c(state)
                    cls = (time.perf_counter() - t0) / n_c
                    extra["end_to_end_components"] = {"value": 128 * (N // 128) / cls, "unit": "columns/s", "ms_per_step": cls * 1e3, "calls": n_c,
                                                      "note": "RRTMGShortwave()(state) + RRTMGLongwave()(state) on get_default_state(128 x %d x %d): "
                                                              "sympl-style DataArrays in, tendencies + diagnostics out" % (N // 128, L)}
                except Exception as e:   # pragma: no cover
                    extra["end_to_end_components"] = {"error": repr(e)[:200]}
                # (c') the same classes with McICA and the reference's DEFAULT generator (Mersenne twister), clouds in ten layers
                if not cloudy:
                    try:
                        sw_m = climt_amd.RRTMGShortwave(mcica=True, cloud_overlap_method="maximum_random")
                        lw_m = climt_amd.RRTMGLongwave(mcica=True, cloud_overlap_method="maximum_random", allow_synthetic_tables=True)
                        st_m = climt_amd.get_default_state([sw_m, lw_m], grid_state=climt_amd.get_grid(nx=128, ny=N // 128, nz=L))
                        st_m["cloud_area_fraction_in_atmosphere_layer"].values[L // 3:L // 2] = 0.4
                        st_m["mass_content_of_cloud_liquid_water_in_atmosphere_layer"].values[L // 3:L // 2] = 0.03
                        sw_m(st_m); lw_m(st_m)      # (the first call of a grid shape builds the generator's jump polynomials: ~0.5 s, once)
                        t0 = time.perf_counter()
                        for _ in range(n_c):
                            sw_m(st_m); lw_m(st_m)
                        clm = (time.perf_counter() - t0) / n_c
                        extra["end_to_end_components_mcica"] = {"value": 128 * (N // 128) / clm, "unit": "columns/s", "ms_per_step": clm * 1e3, "calls": n_c,
                                                                "note": "RRTMGShortwave(mcica=True) + RRTMGLongwave(mcica=True), maximum-random overlap, the reference's default "
                                                                        "random_number_generator (mersenne_twister: its one sequential stream is built on the device by jump-ahead), "
                                                                        "cloud fraction 0.4 in %d layers of get_default_state(128 x %d x %d)" % (L // 2 - L // 3, N // 128, L)}
                    except Exception as e:   # pragma: no cover
                        extra["end_to_end_components_mcica"] = {"error": repr(e)[:200]}
                # (d) SURVEY 8(f)3: the whole model step either side of the path with the state resident on the device --
                # Instellation -> RRTMG SW + LW (every step) -> Adams-Bashforth -> SlabSurface on a DeviceState, through the
                # same component classes; the host does not wait between steps
                try:
                    import datetime as dtm
                    import climt_amd
                    kw = dict(mcica=True, random_number_generator="kissvec") if cloudy else {}
                    sun, slab = climt_amd.Instellation(), climt_amd.SlabSurface()
                    lw_d, sw_d = climt_amd.RRTMGLongwave(allow_synthetic_tables=True, **kw), climt_amd.RRTMGShortwave(**kw)
                    host_state = climt_amd.get_default_state([sun, lw_d, sw_d, slab], grid_state=climt_amd.get_grid(nx=128, ny=N // 128, nz=L))
                    ds = climt_amd.DeviceState.from_host(host_state, [sun, lw_d, sw_d, slab])
                    stepper = climt_amd.DeviceAdamsBashforth(lw_d, sw_d, slab, wait_every_step=False)
                    dt_model = dtm.timedelta(seconds=600)

                    def model_step():
                        nonlocal ds
                        ds.update(sun(ds))
                        diag, ds = stepper(ds, dt_model)
                        ds.update(diag)
                        ds["time"] = ds["time"] + dt_model
                    for _ in range(3):
                        model_step()
                    _hip.synchronize()
                    n_m = 300
                    t0 = time.perf_counter()
                    for _ in range(n_m):
                        model_step()
                    ds.ctx.synchronize()
                    _hip.synchronize()
                    mstep = (time.perf_counter() - t0) / n_m
                    extra["device_resident_model_step"] = {
                        "value": 128 * (N // 128) / mstep, "unit": "columns/s", "ms_per_step": mstep * 1e3, "steps": n_m,
                        "note": "Instellation -> RRTMGShortwave + RRTMGLongwave (refreshed every step) -> DeviceAdamsBashforth -> SlabSurface on "
                                "climt_amd.DeviceState(get_default_state(128 x %d x %d)): the state stays in HBM, tendency sums and the time "
                                "step are kernels, nothing returns to the host (the default state has no clouds)" % (N // 128, L)}
                except Exception as e:   # pragma: no cover
                    extra["device_resident_model_step"] = {"error": repr(e)[:200]}
                res["extra"] = extra
            if not a.no_cpu_baseline:
                res["cpu_baseline"] = cpu_baseline(L, cloudy)
    # The JSON line must be the LAST line on stdout: RCCL (NCCL_DEBUG=VERSION) writes its banner through C stdio, which
    # would otherwise be flushed after it at exit.  Everyone flushes C stdio, the ranks meet, then rank 0 prints.
    if multi:
        import ctypes
        sys.stdout.flush()
        ctypes.CDLL(None).fflush(None)
        dist.barrier()
        try:
            comm.close()
        except Exception:
            pass
        dist.destroy_process_group()
        ctypes.CDLL(None).fflush(None)
    if res is not None:
        print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
