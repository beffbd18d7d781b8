"""Column sharding of the RRTMG path across GPUs: one process per GPU, RCCL over xGMI for the output gather.

Every routine of the path is column-independent (rrtmg_lw_rad.nomcica.f90:453 and rrtmg_sw_rad.nomcica.f90:587
are serial loops over independent columns), so ranks own contiguous column blocks, the tables are replicated and nothing is
exchanged while computing.  The only communication reassembles the outputs, and it is optional:

    gather = "all"   one ncclAllGather of a flat device buffer holding the rank's 12 (14 with dF/dT) output arrays
             "direct" the same result by one GROUPED ncclSend / ncclRecv exchange: every rank sends its block straight to each
                     of its world-1 peers and receives theirs.  xGMI is point to point (7 links per GPU on an 8-GPU node): a
                     ring all-gather moves (world-1) blocks over ONE link per GPU in world-1 dependent steps, the direct
                     exchange puts one block on each of the 7 links at once.  The own block is not copied at
                     all: readers take it from the rank's local buffer
             "root"  the blocks are sent to rank 0 only (grouped ncclSend / ncclRecv)
             "none"  every rank keeps its block (a model that is itself domain-decomposed needs nothing else)
    unpack = True    the gathered buffer -- [rank][array][level][local column], the collective's layout -- is also written out
                     in the boundary layout [array][level][column] by a block-copy kernel behind the gather, on the same
                     stream: a device consumer gets what a single-GPU call would have produced (gathered_device)

`RcclComm` binds librccl.so directly (ctypes: ncclGetUniqueId / ncclCommInitRank / ncclAllGather / ncclSend / ncclRecv /
ncclGroupStart / ncclGroupEnd) and
runs on its own HIP stream, which is made to wait for the radiation kernels on the device (rrtmg_hip_stream_wait): the gather
of step i runs under the kernels of step i+1, into the other half of a double buffer.  Any object with the same five methods
can stand in for it: the world-size-2 CPU tests use one over torch.distributed's gloo (tests/torch_comm.py) -- this package
itself imports no torch.

Sharded == unsharded, bit for bit, for ANY number of columns and ranks: kissvec sub-columns are seeded per column; for the
Mersenne twister, whose reference stream is ONE sequence over (sub-column, column, layer), every rank passes its block's
position (shard_col0, shard_ncol) and the generator starts at the rank's own draws (jump-ahead, csrc/rrtmg_mt_device.hip); and
block boundaries are multiples of the 64-column tile (column_block(align=64)), so that every column runs in the same tile --
and therefore in the same solve-kernel variant (clear-sky / cloudy tile) -- as in the unsharded call.  (A boundary inside a
tile could move a column to the other variant, which changes the shortwave by round-off.)
"""
import ctypes as C
import os
import socket
import struct

import numpy as np

from ._lib import LW_OUT, SW_OUT

# column axis of every boundary-level array (None: not a per-column array) -- explicit, never guessed from shapes
COLUMN_AXIS = dict(
    play=1, plev=1, tlay=1, tlev=1, h2o=1, o3=1, co2=1, ch4=1, n2o=1, o2=1, cfc11=1, cfc12=1, cfc22=1, ccl4=1,
    cldfr=1, cicewp=1, cliqwp=1, reice=1, reliq=1,                       # [layer][column]
    tsfc=0, asdir=0, asdif=0, aldir=0, aldif=0, coszen=0, lat=0,          # [column]
    emis=1,                                                               # [band][column]
    taucld=1, ssacld=1, asmcld=1, fsfcld=1, cldfmcl=1,                    # [layer][column][band | g-point]
    tauaer=2, ssaaer=2, asmaer=2, ecaer=2,                                # [band | type][layer][column]
    bndsolvar=None, indsolvar=None,
)


GATHER_MODES = ("all", "direct", "root", "none")
TILE = 64   # columns of a wavefront tile (csrc: one wavefront = 64 columns x one work item)


def column_block(ncol, world, rank, align=TILE):
    """Contiguous block [lo, hi) of rank `rank`.  Boundaries are multiples of `align` columns (the last block ends at ncol):
    the tiles are dealt out, blocks differ by at most one tile -- a rank may get nothing when there are fewer tiles than ranks.
    align=1: blocks that differ by at most one column."""
    ntile = -(-ncol // align)
    base, rem = divmod(ntile, world)
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return min(ncol, lo * align), min(ncol, hi * align)


def slice_columns(inp, lo, hi):
    """Columns [lo, hi) of a boundary-level input dict, by the explicit per-name axis map."""
    out = {}
    for k, v in inp.items():
        if not isinstance(v, np.ndarray):
            out[k] = v
            continue
        if k not in COLUMN_AXIS:
            raise KeyError("slice_columns: no column axis known for array '%s'" % k)
        ax = COLUMN_AXIS[k]
        if ax is None:
            out[k] = v
        else:
            idx = [slice(None)] * v.ndim
            idx[ax] = slice(lo, hi)
            out[k] = np.ascontiguousarray(v[tuple(idx)])
    return out


# ---- communicators ---------------------------------------------------------------------------------------------
def tcp_broadcast(payload, rank, world, addr=None, port=None, timeout=120.0):
    """Rank 0's `payload` (bytes) on every rank: a minimal rendezvous over TCP (MASTER_ADDR, RRTMG_HIP_RDZV_PORT or
    MASTER_PORT + 17) used to hand out the RCCL unique id without torch."""
    if world == 1:
        return payload
    addr = addr or os.environ.get("MASTER_ADDR", "127.0.0.1")
    port = int(port or os.environ.get("RRTMG_HIP_RDZV_PORT", int(os.environ.get("MASTER_PORT", "29500")) + 17))
    if rank == 0:
        srv = socket.socket()
        srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        srv.bind((addr, port))
        srv.listen(world)
        srv.settimeout(timeout)
        for _ in range(world - 1):
            conn, _ = srv.accept()
            conn.sendall(struct.pack("<I", len(payload)) + payload)
            conn.close()
        srv.close()
        return payload
    import time
    t0 = time.time()
    while True:
        try:
            s = socket.create_connection((addr, port), timeout=5.0)
            break
        except OSError:
            if time.time() - t0 > timeout:
                raise
            time.sleep(0.05)
    def rd(n):
        b = b""
        while len(b) < n:
            chunk = s.recv(n - len(b))
            if not chunk:
                raise ConnectionError("rendezvous connection closed")
            b += chunk
        return b
    n = struct.unpack("<I", rd(4))[0]
    data = rd(n)
    s.close()
    return data


class _NcclUniqueId(C.Structure):
    _fields_ = [("internal", C.c_char * 128)]


class RcclError(RuntimeError):
    pass


class RcclComm:
    """librccl.so through ctypes: the communicator, its HIP stream, all-gather and gather-to-root of fp64 buffers."""
    NCCL_FLOAT64 = 8   # ncclDataType_t: ncclFloat64 / ncclDouble

    def __init__(self, rank, world, device, broadcast=None):
        from . import _hip
        self.rank, self.world = rank, world
        self.lib = None
        for name in (os.environ.get("RRTMG_HIP_RCCL_LIB"), "librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"):
            if not name:
                continue
            try:
                self.lib = C.CDLL(name)
                break
            except OSError:
                continue
        if self.lib is None:
            raise RcclError("librccl.so not loadable")
        L = self.lib
        L.ncclGetErrorString.restype = C.c_char_p
        L.ncclGetUniqueId.argtypes = [C.POINTER(_NcclUniqueId)]
        L.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, _NcclUniqueId, C.c_int]
        L.ncclAllGather.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p]
        L.ncclSend.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.ncclRecv.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.ncclCommDestroy.argtypes = [C.c_void_p]
        L.ncclGroupStart.argtypes = []
        L.ncclGroupEnd.argtypes = []
        _hip.set_device(device)
        uid = _NcclUniqueId()
        if rank == 0:
            self._ck(L.ncclGetUniqueId(C.byref(uid)), "ncclGetUniqueId")
        if world > 1:      # (string_at: a c_char array field would stop at the first NUL byte)
            raw = (broadcast or tcp_broadcast)(C.string_at(C.addressof(uid), 128) if rank == 0 else b"", rank, world)
            if len(raw) != 128:
                raise RcclError("unique-id exchange returned %d bytes" % len(raw))
            C.memmove(C.addressof(uid), raw, 128)
        self.comm = C.c_void_p()
        self._ck(L.ncclCommInitRank(C.byref(self.comm), world, uid, rank), "ncclCommInitRank")
        self.stream = _hip.Stream()
        self.kind = "rccl"

    def _ck(self, rc, what):
        if rc != 0:
            raise RcclError("%s failed: %s" % (what, self.lib.ncclGetErrorString(rc).decode()))

    def all_gather(self, send_ptr, recv_ptr, count):
        """recv[r * count : (r+1) * count] = rank r's send[0:count] (fp64), on the communicator's stream."""
        self._ck(self.lib.ncclAllGather(send_ptr, recv_ptr, count, self.NCCL_FLOAT64, self.comm, self.stream.s), "ncclAllGather")

    def gather_root(self, send_ptr, recv_ptr, count):
        """Rank 0 receives every other rank's block into recv[r * count ...]; its own block stays where it is."""
        L = self.lib
        self._ck(L.ncclGroupStart(), "ncclGroupStart")
        if self.rank == 0:
            for r in range(1, self.world):
                self._ck(L.ncclRecv(recv_ptr + 8 * count * r, count, self.NCCL_FLOAT64, r, self.comm, self.stream.s), "ncclRecv")
        else:
            self._ck(L.ncclSend(send_ptr, count, self.NCCL_FLOAT64, 0, self.comm, self.stream.s), "ncclSend")
        self._ck(L.ncclGroupEnd(), "ncclGroupEnd")

    def exchange_direct(self, send_ptr, recv_ptr, count):
        """Every rank r != me receives my send[0:count] and I receive theirs into recv[r * count ...], as ONE group of 2 (world-1)
        point-to-point operations (RCCL runs the pairs concurrently, each over the xGMI link between the two GPUs);
        recv[me * count ...] is not written -- the own block stays in `send`."""
        L = self.lib
        self._ck(L.ncclGroupStart(), "ncclGroupStart")
        for d in range(1, self.world):
            to, frm = (self.rank + d) % self.world, (self.rank - d) % self.world      # (rotated: no two ranks start on the same peer)
            self._ck(L.ncclSend(send_ptr, count, self.NCCL_FLOAT64, to, self.comm, self.stream.s), "ncclSend")
            self._ck(L.ncclRecv(recv_ptr + 8 * count * frm, count, self.NCCL_FLOAT64, frm, self.comm, self.stream.s), "ncclRecv")
        self._ck(L.ncclGroupEnd(), "ncclGroupEnd")

    def wait(self):
        self.stream.synchronize()

    def close(self):
        if getattr(self, "comm", None):
            self.lib.ncclCommDestroy(self.comm)
            self.comm = None


def comm_selftest(comm, alloc=None, words=128, check_untouched=True):
    """First-execution insurance for a communicator at world > 1: every gather mode once on a small pattern -- word i of
    rank r's block is r * 4096 + i + 0.25, so a block that lands in the wrong slot, arrives partly, or is never written shows
    up as such -- checked on THIS rank.  -> {"all" | "direct" | "root": "OK" | "FAIL: ..."} (a mode whose call raises reports
    the exception; nothing propagates: the caller goes on measuring and prints which collective misbehaved).

    `comm` is an RcclComm or anything with its five methods; `alloc(shape)` returns a buffer with .ptr / .download() and an
    upload from numpy (climt_amd._hip.DeviceArray by default); alloc="host": numpy arrays are handed to the communicator
    (the gloo communicator of the CPU tests).  check_untouched=False: blocks a mode has no business writing are not looked at
    (the testing communicator over torch.distributed gathers the own block too, which readers of `direct` never look at)."""
    rank, world = comm.rank, comm.world
    host = alloc == "host"
    if alloc is None:
        from . import _hip
        alloc = _hip.DeviceArray

    def pattern(r):
        return r * 4096.0 + np.arange(words, dtype=np.float64) + 0.25
    expect = np.concatenate([pattern(r) for r in range(world)])
    out = {}
    for mode in ("all", "direct", "root"):
        try:
            send_h, recv_h = pattern(rank), np.full(words * world, -1.0)
            if host:
                send, recv = send_h, recv_h
                sp, rp = send, recv
            else:
                send, recv = alloc((words,)), alloc((words * world,))
                send.upload(send_h); recv.upload(recv_h)
                sp, rp = send.ptr, recv.ptr
            {"all": comm.all_gather, "direct": comm.exchange_direct, "root": comm.gather_root}[mode](sp, rp, words)
            comm.wait()
            got = recv if host else recv.download()
            got = np.asarray(got, dtype=np.float64).reshape(world, words)
            want = expect.reshape(world, words)
            bad = []
            for r in range(world):
                written = mode == "all" or (mode == "direct" and r != rank) or (mode == "root" and rank == 0 and r != 0)
                if written and not np.array_equal(got[r], want[r]):
                    k = int(np.flatnonzero(got[r] != want[r])[0])
                    bad.append("block %d word %d: %r != %r (%d of %d words differ)" % (r, k, float(got[r][k]), float(want[r][k]),
                                                                                        int((got[r] != want[r]).sum()), words))
                elif check_untouched and not written and not np.all(got[r] == -1.0):
                    bad.append("block %d must stay untouched in mode %s and was written" % (r, mode))
            out[mode] = "OK" if not bad else "FAIL on rank %d: %s" % (rank, "; ".join(bad[:3]))
        except Exception as e:      # noqa: BLE001 -- the point is to report, per mode, what happened
            out[mode] = "FAIL on rank %d: %s: %s" % (rank, type(e).__name__, str(e)[:200])
    return out


# ---- the sharded radiation step ----------------------------------------------------------------------------------
class ShardedRadiation:
    """This rank's block of a column grid, resident on its GPU, and the (double-buffered) gather of the outputs.

        sr = ShardedRadiation(ctx, comm, ncol_total, nlay, gather="all")
        sr.set_inputs(full_inputs)          # slices this rank's columns and uploads them once
        for ...: b = sr.step(mcica=True)    # SW || LW on the context's two streams, then the gather of buffer b
        sr.finish(); full = sr.gathered_host(b)

    `ctx` is a climt_amd._lib.Context (device memory) -- or the tests' host emulation, in which case the buffers are numpy.
    """

    def __init__(self, ctx, comm, ncol_total, nlay, gather="all", idrv=False, device=True, nbuf=2, allocator=None, force=False, unpack=False,
                 align=TILE):
        """unpack=True: on the ranks that hold the gathered outputs they are also put into the BOUNDARY layout -- every array
        [levels][ncol_total], column fastest, as a single-GPU call would have written it (rrtmg_lw_c_binder.f90:198-202) -- by
        a block-copy kernel behind the gather on the communicator's stream (rrtmg_hip_copy_blocks): gathered_device(b) /
        gathered_host(b).  Without it the gathered buffer keeps the collective's layout [rank][array][level][local column]
        and only gathered_host reassembles it."""
        if gather not in GATHER_MODES:
            raise ValueError("gather must be one of %s" % (GATHER_MODES,))
        self.ctx, self.comm, self.gather, self.device = ctx, comm, gather, device
        self.rank, self.world = comm.rank, comm.world
        self.ncol_total, self.nlay = ncol_total, nlay
        self.align = align
        self.lo, self.hi = column_block(ncol_total, self.world, self.rank, align)
        self.ncol = self.hi - self.lo
        # widest block: the per-rank stride of the gathered buffer
        self.width = max(hi - lo for lo, hi in (column_block(ncol_total, self.world, r, align) for r in range(self.world)))
        self.names = [k for k, _ in SW_OUT] + [k for k, _ in LW_OUT] + (["duflx_dt", "duflxc_dt"] if idrv else [])
        self.levs = [lev for _, lev in SW_OUT] + [lev for _, lev in LW_OUT] + ([1, 1] if idrv else [])
        self.idrv = idrv
        self.block = sum((nlay + lev) * self.width for lev in self.levs)       # doubles per rank in the gathered buffer
        self.do_gather = gather != "none" and (self.world > 1 or force)     # force: run the collective with one rank too (tests)
        self.nbuf = nbuf if self.do_gather else 1
        gathered_here = self.do_gather and (gather in ("all", "direct") or (gather == "root" and self.rank == 0))
        if device:
            from . import _hip
            self._hip = _hip
            alloc = allocator or (lambda shape: _hip.DeviceArray(shape))     # (a communicator may need to own the buffers)
            self.flat = [alloc((self.block,)) for _ in range(self.nbuf)]
            self.full = [alloc((self.block * self.world,)) if gathered_here else None for _ in range(self.nbuf)]
            for buf in self.flat:      # a rank with fewer columns than the widest block (or none) sends the padding too: defined bytes
                if hasattr(buf, "zero"):
                    buf.zero()
            self.events = [_hip.Event() for _ in range(self.nbuf)]
            self._prev_deferred = ctx.set_deferred(True)
        else:
            self.flat = [np.zeros(self.block) for _ in range(self.nbuf)]
            self.full = [np.zeros(self.block * self.world) if gathered_here else None for _ in range(self.nbuf)]
        self.inflight = [False] * self.nbuf
        self.i = 0
        self.inp = None
        self._keep = None
        self.unpack = bool(unpack) and gathered_here
        if self.unpack:
            self._setup_unpack(allocator)

    # ---- boundary layout of the gathered outputs ------------------------------------------------------------------
    def boundary_offsets(self):
        """name -> (offset in doubles, levels) of each output array in the unpacked buffer: the arrays one after the other,
        each [levels][ncol_total]."""
        off, out = 0, {}
        for k, lev in zip(self.names, self.levs):
            out[k] = (off, self.nlay + lev)
            off += (self.nlay + lev) * self.ncol_total
        return out

    def unpack_descriptors(self):
        """One block copy per (rank, array): (src_off, dst_off, rows, cols, src_stride, dst_stride, from_own_flat) in doubles.
        src: the gathered buffer [rank][array][level][that rank's columns]; with gather='root' (rank 0) and gather='direct'
        (every rank) the own block never enters it and is read from the rank's local buffer instead (from_own_flat)."""
        dst = self.boundary_offsets()
        out = []
        for r in range(self.world):
            rlo, rhi = column_block(self.ncol_total, self.world, r, self.align)
            n_r = rhi - rlo
            own = self.gather in ("root", "direct") and r == self.rank
            for k, (off, rows) in self.offsets(n_r).items():
                out.append(((0 if own else r * self.block) + off, dst[k][0] + rlo, rows, n_r, n_r, self.ncol_total, own))
        return out

    def _setup_unpack(self, allocator):
        total = sum(rows * self.ncol_total for _, rows in self.boundary_offsets().values())
        desc = self.unpack_descriptors()
        self._desc = desc
        if self.device:
            alloc = allocator or (lambda shape: self._hip.DeviceArray(shape))
            self.unpacked = [alloc((total,)) for _ in range(self.nbuf)]
            self._desc_dev = []
            for own in (False, True):      # two launches at most: blocks read from the gathered buffer / from the own local one
                rows = [d[:6] for d in desc if d[6] == own]
                if rows:
                    arr = np.ascontiguousarray(rows, dtype=np.int64)
                    self._desc_dev.append((own, self._hip.DeviceArray.from_host(arr), len(rows), max(d[2] for d in rows), max(d[3] for d in rows)))
        else:
            self.unpacked = [np.zeros(total) for _ in range(self.nbuf)]

    def _run_unpack(self, b):
        """Behind the gather of buffer b, in stream order on the communicator's stream (host arrays: at once)."""
        if not self.device:
            full, flat, out = self.full[b], self.flat[b], self.unpacked[b]
            for so, do, rows, cols, ss, ds, own in self._desc:
                src = flat if own else full
                for r in range(rows):
                    out[do + r * ds: do + r * ds + cols] = src[so + r * ss: so + r * ss + cols]
            return
        stream = getattr(getattr(self.comm, "stream", None), "s", None)
        if stream is None:
            self.comm.wait()       # a communicator without a stream of its own: the gather is complete before the copy starts
        for own, d, n, mr, mc in self._desc_dev:
            src = self.flat[b].ptr if own else self.full[b].ptr
            self.ctx.copy_blocks(d.ptr, n, mr, mc, src, self.unpacked[b].ptr, stream=stream)

    def gathered_device(self, b):
        """name -> (device pointer, (levels, ncol_total)) of the full-grid outputs of buffer b in the boundary layout
        (needs unpack=True; valid once the gather of buffer b has completed: finish(), or the buffer's event)."""
        if not self.unpack:
            raise RuntimeError("ShardedRadiation(unpack=True) on a rank that holds the gathered outputs is needed for gathered_device")
        base = self.unpacked[b].ptr if self.device else self.unpacked[b]
        if self.device:
            return {k: (base + 8 * off, (rows, self.ncol_total)) for k, (off, rows) in self.boundary_offsets().items()}
        return {k: (base[off:off + rows * self.ncol_total].reshape(rows, self.ncol_total), (rows, self.ncol_total)) for k, (off, rows) in self.boundary_offsets().items()}

    # layout of one rank's block: the arrays one after the other, each [levels][that rank's columns]
    def offsets(self, ncol):
        off, out = 0, {}
        for k, lev in zip(self.names, self.levs):
            out[k] = (off, self.nlay + lev)
            off += (self.nlay + lev) * ncol
        return out

    def set_inputs(self, inp, already_local=False):
        """Boundary-level inputs: the full grid (sliced here) or, with already_local, this rank's block."""
        local = dict(inp) if already_local else slice_columns(inp, self.lo, self.hi)
        local.pop("lat", None)
        local.update(shard_col0=self.lo, shard_ncol=self.ncol_total)
        if self.device:
            self._keep = {k: self._hip.DeviceArray.from_host(v) for k, v in local.items() if isinstance(v, np.ndarray)}
            self.inp = {k: v.ptr for k, v in self._keep.items()}
            self.inp.update({k: v for k, v in local.items() if not isinstance(v, np.ndarray)})
            self.inp.update(ncol=self.ncol, nlay=self.nlay)
        else:
            self.inp = local

    def _out(self, b):
        offs = self.offsets(self.ncol)
        if self.device:
            base = self.flat[b].ptr
            o = {k: base + 8 * off for k, (off, _) in offs.items()}
        else:
            o = {k: self.flat[b][off:off + n * self.ncol].reshape(n, self.ncol) for k, (off, n) in offs.items()}
        sw = {k: o[k] for k, _ in SW_OUT}
        lw = {k: o[k] for k in self.names[len(SW_OUT):]}
        return sw, lw

    def step(self, mcica=False, host_wait=False, sync=True):
        """One LW+SW pass over this rank's block into buffer b = step number mod nbuf; starts its gather; returns b.
        host_wait: the communicator cannot be ordered after the kernels on the device (no stream of its own): the host
        waits for the kernels before it starts the gather.
        sync=False: do not wait for this step's kernels (a time loop that needs nothing on the host: the streams order
        step i+1 behind step i, the status flags are sticky and are collected by the next synchronizing step or finish();
        the host still waits for the gather that last read the buffer this step writes, nbuf steps back)."""
        b = self.i % self.nbuf
        self.i += 1
        if self.inflight[b]:                      # the gather that read this buffer (nbuf steps ago) must be done
            if self.device and getattr(self, "_host_wait", False):
                self.comm.wait()
            elif self.device:
                self.events[b].synchronize()
            self.inflight[b] = False
        sw, lw = self._out(b)
        ms = 1 if self.device else 0
        if self.ncol > 0:                         # (fewer tiles than ranks: this rank has no columns and only takes part in the gather)
            self.ctx.sw_fluxes(self.inp, mcica=mcica, out=sw, memspace=ms)
            self.ctx.lw_fluxes(self.inp, mcica=mcica, out=lw, memspace=ms)
        if self.do_gather:
            if self.device and host_wait:
                self.ctx.synchronize()
            elif self.device:
                self.ctx.stream_wait(self.comm.stream.s)      # device-side: the gather starts when the kernels are done
            send = self.flat[b].ptr if self.device else self.flat[b]
            recv = (self.full[b].ptr if self.device else self.full[b]) if self.full[b] is not None else None
            if self.gather == "all":
                self.comm.all_gather(send, recv, self.block)
            elif self.gather == "direct":
                self.comm.exchange_direct(send, recv, self.block)
            else:
                self.comm.gather_root(send, recv if recv is not None else 0, self.block)
            if self.unpack:
                self._run_unpack(b)
            if self.device and not host_wait:
                self.events[b].record(self.comm.stream.s)
            self.inflight[b] = True
            self._host_wait = host_wait
        if self.device and sync:
            self.ctx.synchronize()                # this step's kernels are complete and their status checked
        return b

    def finish(self):
        """Wait for every gather in flight and for the kernels (their status is checked)."""
        if self.device:
            self.ctx.synchronize()
        for b in range(self.nbuf):
            if self.inflight[b]:
                if self.device and not getattr(self, "_host_wait", False):
                    self.events[b].synchronize()
                self.inflight[b] = False
        self.comm.wait()

    def close(self):
        """finish() and hand the (shared) context back in the mode it was found in (see DeviceState.close)."""
        self.finish()
        prev = getattr(self, "_prev_deferred", None)
        if self.device and prev is not None:
            self.ctx.set_deferred(prev)
            self._prev_deferred = None

    def gather_ingress_bytes(self):
        """Bytes this rank RECEIVES per step under the current gather mode (what its xGMI links must deliver while the next
        step computes): (world-1) blocks for 'all' / 'direct' and for rank 0 of 'root', nothing otherwise."""
        if not self.do_gather:
            return 0
        if self.gather == "root" and self.rank != 0:
            return 0
        return 8 * self.block * (self.world - 1)

    def local_host(self, b):
        """This rank's outputs of buffer b as numpy arrays [levels][local columns]."""
        flat = self.flat[b].download() if self.device else self.flat[b]
        return {k: flat[off:off + n * self.ncol].reshape(n, self.ncol).copy() for k, (off, n) in self.offsets(self.ncol).items()}

    def gathered_host(self, b):
        """The full-grid outputs of buffer b as numpy arrays (ranks that hold them: all, or rank 0 with gather='root')."""
        # (bytes a gather mode moves over the links, per rank and step: see gather_ingress_bytes)
        if not self.do_gather:
            return self.local_host(b)
        if self.full[b] is None:
            return None
        if self.unpack:        # already in the boundary layout (on the device: one download, no host reassembly)
            flat = self.unpacked[b].download() if self.device else self.unpacked[b]
            return {k: flat[off:off + rows * self.ncol_total].reshape(rows, self.ncol_total).copy() for k, (off, rows) in self.boundary_offsets().items()}
        full = self.full[b].download() if self.device else self.full[b]
        mine = self.local_host(b)
        cols = {k: [] for k in self.names}
        for r in range(self.world):
            rlo, rhi = column_block(self.ncol_total, self.world, r, self.align)
            n_r = rhi - rlo
            for k, (off, n) in self.offsets(n_r).items():
                if r == self.rank and self.gather in ("root", "direct"):
                    cols[k].append(mine[k])     # gather='root' / 'direct' leave the own block in place
                else:
                    cols[k].append(full[r * self.block + off: r * self.block + off + n * n_r].reshape(n, n_r))
        return {k: np.concatenate(v, axis=1) for k, v in cols.items()}
