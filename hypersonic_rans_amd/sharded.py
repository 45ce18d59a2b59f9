"""Chunk-parallel decode of ONE stream over the GPUs of a node (one process per GPU, torch.distributed).

The reference fans independent mt_ blocks out to a thread pool (src/mt_rANS32x64_16w_decode.cpp:182-224).  Here the unit
is a *chain* of the decode plan (an mt_ block, or a checkpoint interval of a raw / block_ stream): the chains are split
into `world_size` contiguous runs balanced by decoded bytes, every rank decodes its run into the same output offsets of
its own buffer, and the disjoint byte ranges are then exchanged — the only collective, and only when the caller wants
the output in one place.

The exchange is point-to-point, like the hardware: on an MI355X node every GPU has a direct xGMI link to each of the
other seven, so rank r sends its range to every peer that wants it with one grouped batch of sends/receives (RCCL
ncclGroupStart{ncclSend/ncclRecv} under torch's batch_isend_irecv): all seven links of a GPU carry data at once and
every byte crosses exactly one link, straight from the decoder's output buffer into the receiver's output buffer — no
staging copy, no padding, no ring.  (A ring all-gather would push 7/8 of the output through each GPU's two ring links.)
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist

from . import api


def shard_chains(plan, world_size: int) -> list[tuple[int, int]]:
    """Splits the plan's chains into `world_size` contiguous runs [(first, count), ...] of (nearly) equal decoded bytes.
    Runs may be empty (count == 0) when there are fewer chains than ranks."""
    hdr, cf, pieces = api.plan_tables(plan)
    n, S = hdr["n_chains"], hdr["states"]
    size = np.where(pieces["flags"] & 2, pieces["fill_len"], pieces["steps"].astype(np.uint64) * S + pieces["tail"]).astype(np.uint64)
    per_piece_end = np.cumsum(size)
    chain_end = per_piece_end[cf[1:].astype(np.int64) - 1]  # decoded bytes up to and including chain c (chains are in output order)
    total = int(chain_end[-1]) if n else 0
    bounds = [0]
    for r in range(1, world_size):
        target = total * r // world_size
        bounds.append(max(bounds[-1], int(np.searchsorted(chain_end, target, side="right"))))
    bounds.append(n)
    return [(bounds[r], bounds[r + 1] - bounds[r]) for r in range(world_size)]


def local_range(plan, first: int, count: int) -> tuple[int, int]:
    return api.plan_chain_range(plan, first, count) if count else (0, 0)


def gather_ranges(out: torch.Tensor, ranges: list[tuple[int, int]], group=None, root: int | None = None) -> torch.Tensor:
    """`out` holds this rank's decoded bytes at their final offsets; `ranges[r]` = [begin, end) owned by rank r.
    After the call `out` is complete on every rank (root None) or on rank `root` only.  In place: every transfer reads
    the owner's slice of its `out` and lands in the same slice of the receiver's `out`; one grouped batch of
    point-to-point operations (see the module docstring)."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if world == 1:
        return out
    ops = []
    b, e = ranges[rank]
    for peer in range(world):
        if peer == rank:
            continue
        g_peer = dist.get_global_rank(group, peer) if group is not None else peer
        if e > b and (root is None or root == peer):
            ops.append(dist.P2POp(dist.isend, out[b:e], g_peer, group))
        pb, pe = ranges[peer]
        if pe > pb and (root is None or root == rank):
            ops.append(dist.P2POp(dist.irecv, out[pb:pe], g_peer, group))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    return out


class ShardedDecoder:
    """Everything that does not change between decodes of one (plan, world) pair, prepared once: this rank's chain run, its
    output range, the stream bytes it needs and its device plan.  `decode` is then one kernel launch + the exchange."""

    def __init__(self, ctx: "api.Context", plan, group=None):
        self.ctx, self.group = ctx, group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.runs = shard_chains(plan, self.world)
        self.ranges = [local_range(plan, f, c) for f, c in self.runs]
        self.total = api.plan_decoded_length(plan)
        hdr, _, _ = api.plan_tables(plan)
        self.stream_len = int(hdr["stream_len"])
        self.first, self.count = self.runs[self.rank]
        self.dplan = None
        self.window = (0, 0)
        self.head = (0, 0)
        if self.count:
            self.dplan = ctx.make_device_plan(api.plan_slice(plan, self.first, self.count))
            (hb, he), (bb, be) = api.plan_stream_ranges(plan, self.first, self.count)
            self.head = (hb, he)
            self.window = (bb & ~15, be)  # 16-byte aligned start: hsrans_decode_device_window

    # -- the stream is already in this rank's HBM, whole --------------------------------------------------------------
    def decode(self, d_stream: torch.Tensor, out: torch.Tensor, gather: bool = True, root: int | None = None) -> torch.Tensor:
        if self.dplan is not None:
            self.ctx.decode_device(self.dplan, d_stream, out, stream_length=self.stream_len)
        return gather_ranges(out, self.ranges, self.group, root) if gather else out

    # -- the stream is in host memory: upload only what this rank's chains read ---------------------------------------
    def upload_window(self, host_stream, device, side_stream: "torch.cuda.Stream | None" = None) -> torch.Tensor:
        """Device copy of this rank's window of the stream (hsrans_plan_stream_ranges body, aligned down to 16 bytes) — the
        buffer is as long as the window, not as the stream.  The copy goes through `side_stream` when given."""
        lo, hi = self.window
        host = host_stream if isinstance(host_stream, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(host_stream, dtype=np.uint8))
        d = torch.empty(max(hi - lo, 16) + 16, dtype=torch.uint8, device=device)
        if hi > lo:
            if side_stream is not None:
                with torch.cuda.stream(side_stream):
                    d[: hi - lo].copy_(host[lo:hi], non_blocking=True)
            else:
                d[: hi - lo].copy_(host[lo:hi], non_blocking=True)
        return d

    def decode_window(self, d_window: torch.Tensor, out: torch.Tensor, gather: bool = True, root: int | None = None) -> torch.Tensor:
        if self.dplan is not None:
            lo, hi = self.window
            self.ctx.decode_device_window(self.dplan, d_window, lo, hi - lo, out)
        return gather_ranges(out, self.ranges, self.group, root) if gather else out

    def global_status(self) -> int:
        """This launch's device status OR-ed over all ranks (one small all-reduce): every rank learns whether any rank's kernel
        met a malformed histogram / block header."""
        st = self.ctx.status(self.dplan) if self.dplan is not None else 0
        dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(self.group) == "nccl" else torch.device("cpu")
        t = torch.tensor([1 if st else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        return int(t.item())

    def check(self) -> None:
        if self.dplan is not None and self.ctx.status(self.dplan) != 0:
            raise api.HsransError("device reported a malformed histogram / block header")


def decode_sharded(ctx: "api.Context", d_stream: torch.Tensor, stream_length: int, plan, gather: bool = True, group=None) -> torch.Tensor:
    """Every rank holds the compressed stream in HBM and the (host) plan; rank r decodes chain run r.
    Returns the full decoded tensor (gather=True) or the local buffer with only this rank's range filled.
    One-shot convenience over ShardedDecoder (which keeps the device plan for repeated decodes)."""
    dec = ShardedDecoder(ctx, plan, group)
    out = torch.empty(dec.total, dtype=torch.uint8, device=d_stream.device)
    dec.decode(d_stream, out, gather=gather)
    dec.check()
    return out


def decode_sharded_from_host(ctx: "api.Context", host_stream, plan, gather: bool = True, group=None) -> torch.Tensor:
    """BASELINE config 4/5 shape: the stream lives in host memory on every rank, each rank uploads only the window its chains
    read (on a side stream), decodes its chains, and the decoded ranges are exchanged."""
    dec = ShardedDecoder(ctx, plan, group)
    dev = torch.device("cuda", torch.cuda.current_device())
    side = torch.cuda.Stream(device=dev)
    d_window = dec.upload_window(host_stream, dev, side)
    out = torch.empty(dec.total, dtype=torch.uint8, device=dev)
    torch.cuda.current_stream(dev).wait_stream(side)
    dec.decode_window(d_window, out, gather=gather)
    dec.check()
    return out
