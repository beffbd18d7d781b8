"""TEST INFRASTRUCTURE: communicators over torch.distributed with the interface of climt_amd.distributed.RcclComm
(all_gather / exchange_direct / gather_root / wait / close, .rank / .world / .kind / .stream), so that climt_amd.distributed.ShardedRadiation can
be exercised where there is no RCCL: world-size-2 gloo on CPU (tests/test_distributed_cpu.py) and, as an explicit option of
bench.py (`--comm torch`), two ranks on one GPU.  The product (climt_amd/) imports no torch."""
import numpy as np


class TorchComm:
    """HOST arrays (numpy) over torch.distributed: gloo in the CPU tests."""

    def __init__(self, dist, rank, world):
        self.dist, self.rank, self.world = dist, rank, world
        self.stream = None
        self.kind = "torch." + dist.get_backend()

    def all_gather(self, send, recv, count):
        import torch
        parts = [torch.empty(count, dtype=torch.float64) for _ in range(self.world)]
        self.dist.all_gather(parts, torch.from_numpy(send[:count]))
        for r, p in enumerate(parts):
            recv[r * count:(r + 1) * count] = p.numpy()

    def gather_root(self, send, recv, count):
        import torch
        t = torch.from_numpy(send[:count])
        if self.rank == 0:
            parts = [torch.empty(count, dtype=torch.float64) for _ in range(self.world)]
            self.dist.gather(t, parts, dst=0)
            for r in range(1, self.world):
                recv[r * count:(r + 1) * count] = parts[r].numpy()
        else:
            self.dist.gather(t, None, dst=0)

    def exchange_direct(self, send, recv, count):
        """point to point, as RcclComm.exchange_direct: the own block is NOT written into recv"""
        import torch
        mine = torch.from_numpy(np.ascontiguousarray(send[:count]))
        got = {}
        reqs = []
        for d in range(1, self.world):
            to, frm = (self.rank + d) % self.world, (self.rank - d) % self.world
            got[frm] = torch.empty(count, dtype=torch.float64)
            reqs.append(self.dist.isend(mine, dst=to))
            reqs.append(self.dist.irecv(got[frm], src=frm))
        for q in reqs:
            q.wait()
        for frm, t in got.items():
            recv[frm * count:(frm + 1) * count] = t.numpy()

    def wait(self):
        pass

    def close(self):
        pass


class _TorchBuf:
    """A device buffer owned by torch: the same .ptr / .download() as climt_amd._hip.DeviceArray."""

    def __init__(self, shape, device):
        import torch
        self.t = torch.empty(int(np.prod(shape)), dtype=torch.float64, device=device)
        self.ptr = self.t.data_ptr()

    def download(self):
        return self.t.cpu().numpy()

    def upload(self, arr):
        self.t.copy_(self.t.new_tensor(np.ascontiguousarray(arr, dtype=np.float64).ravel()))


class TorchDeviceComm:
    """DEVICE buffers that torch allocated, over torch.distributed (backend nccl = RCCL, or gloo with two ranks on one GPU)."""

    def __init__(self, dist, rank, world, device):
        import torch
        self.dist, self.rank, self.world, self.device = dist, rank, world, device
        self.bufs, self.work = {}, []
        self.kind = "torch." + dist.get_backend()

        class _S:      # the library's kernels are waited for on the host with this communicator (no foreign stream handle)
            s = None
        self.stream = _S()
        self.torch = torch

    def alloc(self, shape):
        b = _TorchBuf(shape, self.device)
        self.bufs[b.ptr] = b.t
        return b

    def all_gather(self, send_ptr, recv_ptr, count):
        self.work.append(self.dist.all_gather_into_tensor(self.bufs[recv_ptr], self.bufs[send_ptr][:count], async_op=True))

    def gather_root(self, send_ptr, recv_ptr, count):
        if self.rank == 0:
            parts = list(self.bufs[recv_ptr].view(self.world, count).unbind(0))
            self.work.append(self.dist.gather(self.bufs[send_ptr][:count], parts, dst=0, async_op=True))
        else:
            self.work.append(self.dist.gather(self.bufs[send_ptr][:count], None, dst=0, async_op=True))

    def exchange_direct(self, send_ptr, recv_ptr, count):
        """(testing stand-in: the collective writes the own block too, which `direct` readers never look at)"""
        self.all_gather(send_ptr, recv_ptr, count)

    def wait(self):
        for w in self.work:
            w.wait()
        self.work = []
        self.torch.cuda.synchronize()

    def close(self):
        pass


def sharded_fluxes(ctx, inp, which, mcica, dist, world, rank, align=64):
    """Host arrays: this rank's block of `inp` through ctx.{sw,lw}_fluxes, outputs all-gathered with torch.distributed; returns
    full-size arrays on every rank.  (The device-resident product path is climt_amd.distributed.ShardedRadiation.)"""
    from climt_amd.distributed import column_block, slice_columns
    nlay, ncol = inp["play"].shape
    blocks = [column_block(ncol, world, r, align) for r in range(world)]
    lo, hi = blocks[rank]
    comm = TorchComm(dist, rank, world)
    width = max(b[1] - b[0] for b in blocks)
    if hi > lo:
        local = slice_columns(inp, lo, hi)
        local.update(shard_col0=lo, shard_ncol=ncol)
        out = ctx.sw_fluxes(local, mcica=mcica) if which == "sw" else ctx.lw_fluxes(local, mcica=mcica)
    else:
        from climt_amd._lib import LW_OUT, SW_OUT
        out = {k: np.zeros((nlay + lev, 0)) for k, lev in (SW_OUT if which == "sw" else LW_OUT)}
    full = {}
    for k, v in out.items():
        count = v.shape[0] * width
        send = np.zeros(count)
        send[: v.size] = v.ravel()
        recv = np.zeros(count * world)
        comm.all_gather(send, recv, count)
        cols = []
        for r, (rlo, rhi) in enumerate(blocks):
            cols.append(recv[r * count: r * count + v.shape[0] * (rhi - rlo)].reshape(v.shape[0], rhi - rlo))
        full[k] = np.concatenate(cols, axis=1)
    return full
