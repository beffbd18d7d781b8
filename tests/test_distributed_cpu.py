"""N>1 path on CPU: two processes (torch.distributed, gloo, 127.0.0.1) shard the columns, compute their block
with the host emulation of the device functions, all-gather the outputs, and must reproduce the unsharded
result bit for bit -- what the 8-GPU run relies on."""
import os
import socket
import sys

import numpy as np
import pytest

from helpers import ROOT, EmuContext, load_ref_case


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    from climt_amd.distributed import sharded_fluxes
    from helpers import EmuContext, load_ref_case
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    c, mcica, _ = load_ref_case("mcica_kiss_maxrand")
    ctx = EmuContext()
    sw = sharded_fluxes(ctx, c, "sw", mcica, dist, world, rank)
    lw = sharded_fluxes(ctx, c, "lw", mcica, dist, world, rank)
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, sw, lw))


def test_two_rank_sharding_is_bit_identical():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    c, mcica, _ = load_ref_case("mcica_kiss_maxrand")
    e = EmuContext()
    sw0, lw0 = e.sw_fluxes(c, mcica=mcica), e.lw_fluxes(c, mcica=mcica)
    for rank, sw, lw in res:
        for k in sw0:
            assert np.array_equal(sw[k], sw0[k]), (rank, k)
        for k in lw0:
            assert np.array_equal(lw[k], lw0[k]), (rank, k)


def test_column_blocks_cover_and_balance():
    from climt_amd.distributed import column_block
    for ncol in (1, 7, 64, 8192, 131072, 1036800):
        for world in (1, 2, 3, 8):
            edges = [column_block(ncol, world, r) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == ncol
            assert all(edges[i][1] == edges[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in edges]
            assert max(sizes) - min(sizes) <= 1
