"""N>1 path on CPU: two processes (torch.distributed, gloo, 127.0.0.1) shard the columns, compute their block
with the host emulation of the device functions, gather the outputs, and must reproduce the unsharded
result bit for bit -- what the 8-GPU run relies on.  The product class under test is
climt_amd.distributed.ShardedRadiation (flat double-buffered output buffer, gather modes all / root / none);
on the GPU its communicator is RcclComm (librccl through ctypes), here tests/torch_comm.TorchComm (gloo) on host memory."""
import os
import socket
import sys

import numpy as np
import pytest

from helpers import ROOT, EmuContext, load_ref_case


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _cases():
    """(name, inputs, mcica): kissvec maximum-random, and the Mersenne twister (ONE global stream over (sub-column,
    column, layer): a shard has to start at its own draws -- here the host emulation's skip-ahead, on the device jump-ahead) with maximum and maximum-random overlap."""
    c1, m1, _ = load_ref_case("mcica_kiss_maxrand")
    c2, m2, _ = load_ref_case("mcica_mt_max")
    c3 = dict(c2); c3.update(icld=2, permuteseed=12345)
    return [("kiss_maxrand", c1, m1), ("mt_max", c2, m2), ("mt_maxrand", c3, True)]


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    from climt_amd.distributed import ShardedRadiation
    from helpers import EmuContext
    from torch_comm import TorchComm, sharded_fluxes
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ctx = EmuContext()
    res = {}
    for name, c, mcica in _cases():
        c = {k: v for k, v in c.items() if k != "lat"}
        res[name] = (sharded_fluxes(ctx, c, "sw", mcica, dist, world, rank), sharded_fluxes(ctx, c, "lw", mcica, dist, world, rank))
    # the product class: three steps (both halves of the double buffer are reused), every gather mode
    name, c, mcica = _cases()[0]
    nlay, ncol = c["play"].shape
    for mode in ("all", "direct", "root", "none"):
        sr = ShardedRadiation(ctx, TorchComm(dist, rank, world), ncol, nlay, gather=mode, device=False)
        sr.set_inputs(c)
        for _ in range(3):
            b = sr.step(mcica=mcica)
        sr.finish()
        res["sr_" + mode] = (sr.gathered_host(b), (sr.lo, sr.hi))
    # ... and with the boundary-layout unpack behind the gather (the descriptors the device kernel gets, run by numpy here)
    for mode in ("all", "direct", "root"):
        sr = ShardedRadiation(ctx, TorchComm(dist, rank, world), ncol, nlay, gather=mode, device=False, unpack=True)
        sr.set_inputs(c)
        for _ in range(3):
            b = sr.step(mcica=mcica)
        sr.finish()
        dev = {k: v[0].copy() for k, v in sr.gathered_device(b).items()} if sr.unpack else None
        res["un_" + mode] = (sr.gathered_host(b), dev, sr.unpack_descriptors())
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, res))


def test_two_rank_sharding_is_bit_identical():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=600) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    e = EmuContext()
    whole = {}
    for name, c, mcica in _cases():
        sw0, lw0 = e.sw_fluxes(c, mcica=mcica), e.lw_fluxes(c, mcica=mcica)
        whole[name] = (sw0, lw0)
        for rank in (0, 1):
            sw, lw = res[rank][name]
            for k in sw0:
                assert np.array_equal(sw[k], sw0[k]), (name, rank, k)
            for k in lw0:
                assert np.array_equal(lw[k], lw0[k]), (name, rank, k)
    sw0, lw0 = whole["kiss_maxrand"]
    full = dict(sw0); full.update(lw0)
    for rank in (0, 1):
        for mode in ("all", "direct"):      # the two algorithms of the same gather: ncclAllGather / grouped send-recv
            got, _ = res[rank]["sr_" + mode]
            assert set(got) == set(full)
            assert all(np.array_equal(got[k], full[k]) for k in full), (rank, mode)
        got, (lo, hi) = res[rank]["sr_none"]
        assert all(np.array_equal(got[k], full[k][:, lo:hi]) for k in full), rank
        got, _ = res[rank]["sr_root"]
        if rank == 0:
            assert all(np.array_equal(got[k], full[k]) for k in full)
        else:
            assert got is None
        # unpacked: every array [levels][all columns], on every rank that holds the gather
        for mode in ("all", "direct", "root"):
            got, dev, desc = res[rank]["un_" + mode]
            if mode == "root" and rank != 0:
                assert got is None and dev is None
                continue
            assert all(np.array_equal(got[k], full[k]) and np.array_equal(dev[k], full[k]) for k in full), (rank, mode)
            assert len(desc) == 2 * len(full) and sum(d[2] * d[3] for d in desc) == sum(v.size for v in full.values())
            assert [d[6] for d in desc].count(True) == (0 if mode == "all" else len(full))      # root / direct: own block from the local buffer


def test_mersenne_twister_shards_differ_without_the_skip_ahead():
    """The defect the skip-ahead repairs: a shard that restarts the global stream gets other bits (and correlated draws)."""
    from climt_amd.distributed import slice_columns
    c, mcica, _ = load_ref_case("mcica_mt_max")
    c = {k: v for k, v in c.items() if k != "lat"}
    e = EmuContext()
    whole = e.sw_fluxes(c, mcica=True)
    ncol = c["play"].shape[1]
    sub = slice_columns(c, 10, ncol)
    naive = e.sw_fluxes(sub, mcica=True)
    assert not np.array_equal(naive["swdflx"], whole["swdflx"][:, 10:])
    sub.update(shard_col0=10, shard_ncol=ncol)
    right = e.sw_fluxes(sub, mcica=True)
    assert all(np.array_equal(right[k], whole[k][:, 10:]) for k in whole)


def test_slice_columns_uses_the_axis_map_not_shapes():
    """Column counts equal to a band count (14, 16), to the aerosol-type count (6) or to len(indsolvar) (2) must not
    change which axis is cut."""
    from climt_amd.distributed import slice_columns
    for ncol in (2, 6, 14, 16):
        nlay = ncol                       # square arrays too
        inp = dict(play=np.arange(nlay * ncol, dtype=float).reshape(nlay, ncol), tsfc=np.arange(ncol, dtype=float),
                   emis=np.arange(16 * ncol, dtype=float).reshape(16, ncol),
                   taucld=np.arange(nlay * ncol * 14, dtype=float).reshape(nlay, ncol, 14),
                   tauaer=np.arange(14 * nlay * ncol, dtype=float).reshape(14, nlay, ncol),
                   ecaer=np.arange(6 * nlay * ncol, dtype=float).reshape(6, nlay, ncol),
                   bndsolvar=np.arange(16, dtype=float), indsolvar=np.arange(2, dtype=float), icld=1)
        out = slice_columns(inp, 1, 2)
        assert out["play"].shape == (nlay, 1) and np.array_equal(out["play"][:, 0], inp["play"][:, 1])
        assert out["tsfc"].shape == (1,) and out["emis"].shape == (16, 1)
        assert out["taucld"].shape == (nlay, 1, 14) and np.array_equal(out["taucld"][:, 0, :], inp["taucld"][:, 1, :])
        assert out["tauaer"].shape == (14, nlay, 1) and out["ecaer"].shape == (6, nlay, 1)
        assert out["bndsolvar"].shape == (16,) and out["indsolvar"].shape == (2,) and out["icld"] == 1
    with pytest.raises(KeyError):
        slice_columns(dict(mystery=np.zeros((3, 3))), 0, 1)


def test_column_blocks_cover_and_balance():
    """Blocks are contiguous, cover the grid, start on tile boundaries (so that a column runs in the same 64-column tile --
    the same solve-kernel variant -- sharded or not: bitwise shard == whole for ANY column count) and differ by at most a tile."""
    from climt_amd.distributed import column_block
    for ncol in (1, 7, 64, 100, 1000, 8192, 131072, 1036800):
        for world in (1, 2, 3, 8):
            for align in (64, 1):
                edges = [column_block(ncol, world, r, align) for r in range(world)]
                assert edges[0][0] == 0 and edges[-1][1] == ncol
                assert all(edges[i][1] == edges[i + 1][0] for i in range(world - 1))
                assert all(lo % align == 0 for lo, hi in edges if hi > lo)
                sizes = [hi - lo for lo, hi in edges]
                tiles = [-(-n // align) for n in sizes]
                assert max(tiles) - min(tiles) <= 1 and min(sizes) >= 0      # (the last block may also be ragged)
    assert [column_block(1000, 3, r) for r in range(3)] == [(0, 384), (384, 704), (704, 1000)]
    assert [column_block(100, 3, r) for r in range(3)] == [(0, 64), (64, 100), (100, 100)]      # fewer tiles than ranks


def _worker3(rank, world, port, q):
    """1000 columns (not a multiple of anything) over three ranks, one of the blocks ragged; 100 columns: a rank with nothing."""
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    from climt_amd.distributed import ShardedRadiation
    from climt_amd.synthetic import make_columns
    from helpers import EmuContext
    from torch_comm import TorchComm
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ctx = EmuContext()
    res = {}
    for ncol in (1000, 100):
        c = make_columns(ncol, 12, cloudy=True, seed=77); c.pop("lat")
        c.update(icld=2, iaer=0, dyofyr=1, scon=1367.0, isolvar=0, inflg=2, iceflg=1, liqflg=1, irng=1, permuteseed=4242)
        for mode in ("all", "direct"):      # the direct exchange with three ranks: two peers each, rotated; an idle rank takes part
            sr = ShardedRadiation(ctx, TorchComm(dist, rank, world), ncol, 12, gather=mode, device=False, unpack=True)
            sr.set_inputs(c)
            for _ in range(2):
                b = sr.step(mcica=True)
            sr.finish()
            res[(ncol, mode)] = (sr.gathered_host(b), (sr.lo, sr.hi), sr.gather_ingress_bytes())
    # first-execution insurance of the 8-GPU run (bench.py: comm_selftest): every mode on a small pattern, checked on every
    # rank -- here over gloo with three ranks -- and a communicator that misroutes one block is NAMED, per mode
    from climt_amd.distributed import comm_selftest
    res["selftest"] = comm_selftest(TorchComm(dist, rank, world), alloc="host")

    class Misrouting(TorchComm):
        def exchange_direct(self, send, recv, count):
            super().exchange_direct(send, recv, count)
            a, b = [r for r in range(self.world) if r != self.rank][:2]
            tmp = recv[a * count:(a + 1) * count].copy()
            recv[a * count:(a + 1) * count] = recv[b * count:(b + 1) * count]
            recv[b * count:(b + 1) * count] = tmp

        def gather_root(self, send, recv, count):
            raise RuntimeError("no route to rank 0")
    res["selftest_broken"] = comm_selftest(Misrouting(dist, rank, world), alloc="host")
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, res))


def test_three_ranks_any_column_count_is_bit_identical():
    """bit-identical to the 1-GPU result for a general N -- 1000 columns over 3 ranks (blocks of 384 / 320 / 296:
    tile-aligned starts), Mersenne twister with maximum-random overlap, gather + unpack; and 100 columns (2 tiles, rank 2 idle)."""
    import torch.multiprocessing as mp
    from climt_amd.synthetic import make_columns
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker3, args=(r, 3, port, q)) for r in range(3)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=600) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank in range(3):
        assert res[rank]["selftest"] == {"all": "OK", "direct": "OK", "root": "OK"}, res[rank]["selftest"]
        broken = res[rank]["selftest_broken"]
        assert broken["all"] == "OK" and broken["direct"].startswith("FAIL on rank %d: block" % rank) and "no route to rank 0" in broken["root"], broken
    e = EmuContext()
    for ncol, blocks in ((1000, [(0, 384), (384, 704), (704, 1000)]), (100, [(0, 64), (64, 100), (100, 100)])):
        c = make_columns(ncol, 12, cloudy=True, seed=77); c.pop("lat")
        c.update(icld=2, iaer=0, dyofyr=1, scon=1367.0, isolvar=0, inflg=2, iceflg=1, liqflg=1, irng=1, permuteseed=4242)
        full = dict(e.sw_fluxes(c, mcica=True)); full.update(e.lw_fluxes(c, mcica=True))
        for rank in range(3):
            for mode in ("all", "direct"):
                got, block, ingress = res[rank][(ncol, mode)]
                assert block == blocks[rank]
                assert all(np.array_equal(got[k], full[k]) for k in full), (ncol, rank, mode)
                width = max(hi - lo for lo, hi in blocks)
                assert ingress == 2 * 8 * sum((12 + lev) * width for lev in (1, 1, 0, 1, 1, 0) * 2)      # two peers' blocks


def test_tcp_rendezvous_hands_out_rank0s_bytes():
    """The torch-free exchange of the RCCL unique id (128 arbitrary bytes, NULs included)."""
    import multiprocessing as mp
    from climt_amd.distributed import tcp_broadcast
    port = _free_port()
    payload = bytes(range(128))[::-1] + b"\x00" * 0
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rdzv_worker, args=(r, 3, port, payload, q)) for r in range(3)]
    for p in procs:
        p.start()
    got = [q.get(timeout=60) for _ in procs]
    for p in procs:
        p.join(timeout=30)
    assert all(g == payload for g in got)


def _rdzv_worker(rank, world, port, payload, q):
    sys.path.insert(0, ROOT)
    from climt_amd.distributed import tcp_broadcast
    q.put(tcp_broadcast(payload if rank == 0 else b"", rank, world, addr="127.0.0.1", port=port))
