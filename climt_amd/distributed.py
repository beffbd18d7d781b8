"""Column sharding of the RRTMG path across ranks (one process per GPU, torch.distributed).

Every routine of the path is column-independent, so ranks own contiguous column blocks, the
tables are replicated and the only communication is one all-gather that reassembles the output arrays
(RCCL over xGMI with the "nccl" backend on MI355X; "gloo" in the CPU tests).  Sharded == unsharded, bit for bit.
"""
import numpy as np


def column_block(ncol, world, rank):
    """Contiguous block [lo, hi) of rank `rank`; blocks differ by at most one column."""
    base, rem = divmod(ncol, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def slice_columns(inp, lo, hi, ncol):
    """Column slice of a boundary-level input dict (the column axis is the last axis for 1-D/2-D inputs and for the
    [band][layer][column] aerosol arrays, the middle axis for [layer][column][band] cloud optics)."""
    out = {}
    for k, v in inp.items():
        if not isinstance(v, np.ndarray):
            out[k] = v
        elif v.ndim == 3 and v.shape[1] == ncol and v.shape[2] != ncol:
            out[k] = np.ascontiguousarray(v[:, lo:hi, :])
        elif v.shape[-1] == ncol:
            out[k] = np.ascontiguousarray(v[..., lo:hi])
        else:
            out[k] = v
    return out


def sharded_fluxes(ctx, inp, which, mcica, dist, world, rank):
    """Compute this rank's column block with `ctx` and all-gather the outputs; returns full-size arrays on every
    rank.  `dist` is torch.distributed (initialised)."""
    import torch
    nlay, ncol = inp["play"].shape
    lo, hi = column_block(ncol, world, rank)
    local = slice_columns(inp, lo, hi, ncol)
    out = ctx.sw_fluxes(local, mcica=mcica) if which == "sw" else ctx.lw_fluxes(local, mcica=mcica)
    width = -(-ncol // world)
    full = {}
    for k, v in out.items():
        pad = np.zeros((v.shape[0], width))
        pad[:, : hi - lo] = v
        t = torch.from_numpy(pad)
        parts = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(parts, t)
        cols = []
        for r, p in enumerate(parts):
            rlo, rhi = column_block(ncol, world, r)
            cols.append(p.numpy()[:, : rhi - rlo])
        full[k] = np.concatenate(cols, axis=1)
    return full
