"""Chunk-parallel decode of ONE stream over the GPUs of a node (one process per GPU, torch.distributed).

The reference fans independent mt_ blocks out to a thread pool (src/mt_rANS32x64_16w_decode.cpp:182-224).  Here the unit
is a *chain* of the decode plan (an mt_ block, or a checkpoint interval of a raw / block_ stream): the chains are split
into `world_size` contiguous runs balanced by decoded bytes (or by given weights), every rank decodes its run into the
same output offsets of its own buffer, and the disjoint byte ranges are then exchanged — the only collective, and only
when the caller wants the output in one place.

The exchange is point-to-point, like the hardware: on an MI355X node every GPU has a direct xGMI link to each of the
other seven, so rank r sends its range to every peer that wants it with one grouped batch of sends/receives (RCCL
ncclGroupStart{ncclSend/ncclRecv} under torch's batch_isend_irecv): all seven links of a GPU carry data at once and
every byte crosses exactly one link, straight from the decoder's output buffer into the receiver's output buffer — no
staging copy, no padding, no ring.  (A ring all-gather would push 7/8 of the output through each GPU's two ring links.)

Two things keep the links from being the whole story (a GPU decodes 2 TB/s, its seven links take in about half of that):
  * the exchange is PIPELINED behind the decode: every rank's run is cut into `parts` sub-runs; sub-run k's ranges are
    on the links (RCCL's own stream) while sub-run k+1 decodes, so a step costs max(decode, exchange), not the sum;
  * a gather to ONE rank is WEIGHTED: the root does not send, so it takes the share of the chains that makes its own
    decode end together with its last receive (`root_share`); the other ranks split the rest.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist

from . import api


def shard_chains(plan, world_size: int, weights=None) -> list[tuple[int, int]]:
    """Splits the plan's chains into `world_size` contiguous runs [(first, count), ...] of (nearly) equal decoded bytes, or
    of decoded bytes proportional to `weights[r]` (hsrans_shard_layout with one sub-run per rank).  Runs may be empty
    (count == 0) when there are fewer chains than ranks."""
    shards, _ = api.shard_layout(plan, world_size, 1, weights)
    out, nxt = [], 0
    for r in range(world_size):
        f, c, _, _ = shards[r][0]
        out.append((f if c else nxt, c))
        nxt = out[-1][0] + c
    return out


def local_range(plan, first: int, count: int) -> tuple[int, int]:
    return api.plan_chain_range(plan, first, count) if count else (0, 0)


def root_weights(world_size: int, root: int, root_share: float) -> list[float]:
    """Shares for a gather to `root`: the root decodes `root_share` of the bytes (it sends nothing), the others split the rest."""
    root_share = min(max(root_share, 1.0 / world_size), 1.0)
    rest = (1.0 - root_share) / max(world_size - 1, 1)
    return [root_share if r == root else rest for r in range(world_size)]


def balanced_root_share(world_size: int, decode_bytes_per_s: float, inbound_bytes_per_s: float) -> float:
    """The root's share a of the decoded bytes for which its own decode (a / D) takes as long as receiving the rest
    ((1 - a) / B): a = D / (D + B).  D = one GPU's decode rate, B = the root's aggregate inbound rate, both in decoded
    bytes per second (measured: bench.py times an unweighted step first).  Never below the equal share."""
    if world_size <= 1:
        return 1.0
    a = decode_bytes_per_s / (decode_bytes_per_s + inbound_bytes_per_s)
    return min(max(a, 1.0 / world_size), 1.0)


class ShardLayout:
    """Which chains / output bytes / stream bytes each rank owns and how each rank's run is cut into `parts` sub-runs for the
    pipelined exchange: hsrans_shard_layout (pure host arithmetic in the C library, identical on every rank)."""

    def __init__(self, plan, world_size: int, parts: int = 1, weights=None):
        self.world, self.parts = world_size, max(1, parts)
        shards, windows = api.shard_layout(plan, world_size, self.parts, weights)
        self.sub_runs = [[(f, c) for f, c, _, _ in subs] for subs in shards]
        self.sub_ranges = [[(b, e) if c else (0, 0) for _, c, b, e in subs] for subs in shards]
        self.runs, self.ranges = [], []
        for subs in shards:
            live = [x for x in subs if x[1]]
            self.runs.append((live[0][0], sum(x[1] for x in live)) if live else (subs[0][0], 0))
            self.ranges.append((live[0][2], live[-1][3]) if live else (0, 0))
        self.total = api.plan_decoded_length(plan)
        hdr, _, _ = api.plan_tables(plan)
        self.stream_len = int(hdr["stream_len"])
        self.windows = windows


def post_exchange(out: torch.Tensor, ranges: list[tuple[int, int]], group=None, root: int | None = None, out_base: int = 0) -> list:
    """Posts (does not wait for) the point-to-point transfers of one set of ranges: `ranges[r]` = [begin, end) owned by rank r,
    `out[i]` holds output byte out_base + i.  Every rank that owns bytes sends them to every peer that wants them (all peers,
    or `root` only); one grouped batch.  Returns the requests."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if world == 1:
        return []
    ops = []
    b, e = ranges[rank]
    for peer in range(world):
        if peer == rank:
            continue
        g_peer = dist.get_global_rank(group, peer) if group is not None else peer
        if e > b and (root is None or root == peer):
            ops.append(dist.P2POp(dist.isend, out[b - out_base:e - out_base], g_peer, group))
        pb, pe = ranges[peer]
        if pe > pb and (root is None or root == rank):
            ops.append(dist.P2POp(dist.irecv, out[pb - out_base:pe - out_base], g_peer, group))
    return list(dist.batch_isend_irecv(ops)) if ops else []


def gather_ranges(out: torch.Tensor, ranges: list[tuple[int, int]], group=None, root: int | None = None, out_base: int = 0) -> torch.Tensor:
    """`out` holds this rank's decoded bytes at their final offsets; `ranges[r]` = [begin, end) owned by rank r.
    After the call `out` is complete on every rank (root None) or on rank `root` only.  In place: every transfer reads
    the owner's slice of its `out` and lands in the same slice of the receiver's `out`; one grouped batch of
    point-to-point operations (see the module docstring)."""
    for req in post_exchange(out, ranges, group, root, out_base):
        req.wait()
    return out


def pipelined_gather(out: torch.Tensor, layout: ShardLayout, decode_part, group=None, root: int | None = None, out_base: int = 0) -> torch.Tensor:
    """One sharded decode step with the exchange pipelined behind the decode.  `decode_part(k)` launches (GPU: asynchronously
    on the current stream) the decode of this rank's sub-run k into `out`.  Sub-run k's ranges go onto the links as soon as
    its decode is done — the backend's own stream waits for exactly the work queued before the post — while sub-run k+1
    decodes; the requests are waited for together at the end (GPU: the current stream then waits for the transfers).
    A receiving root posts all its receives first: none of them depends on its own decode."""
    rank = dist.get_rank(group)
    reqs = []
    if root is not None and rank == root:
        for k in range(layout.parts):
            reqs += post_exchange(out, [sr[k] for sr in layout.sub_ranges], group, root, out_base)
        for k in range(layout.parts):
            decode_part(k)
    else:
        for k in range(layout.parts):
            decode_part(k)
            reqs += post_exchange(out, [sr[k] for sr in layout.sub_ranges], group, root, out_base)
    for req in reqs:
        req.wait()
    return out


def comm_for(ctx: "api.Context", group=None) -> "api.Comm":
    """The RCCL communicator of the C library (hsrans_comm) for this process's rank in `group`: made once per (context, group) and
    kept by the context; rank 0's hsrans_comm_unique_id travels through the process group — the only thing torch.distributed
    does for the GPU path besides launching the ranks."""
    comms = ctx.__dict__.setdefault("_comms", {})
    key = id(group)
    if key not in comms:
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        ids = [api.Comm.unique_id() if rank == 0 else None]
        if world > 1:
            dist.broadcast_object_list(ids, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        comms[key] = api.Comm(ctx, ids[0], rank, world)
    return comms[key]


class ShardedDecoder:
    """This rank's share of ONE stream decoded by all ranks — a thin wrapper over the C ABI's hsrans_sharded (hsrans_comm.cpp): the
    layout, the sub-runs' device plans, the pipelined point-to-point exchange on RCCL all live in the library; this class only
    holds the tensors' lifetimes and the handful of views tests and bench.py read.

    `parts`   sub-runs per rank for the pipelined exchange (1 = decode, then exchange).
    `weights` decoded-byte shares of the ranks (None = equal; root_weights() for a gather to one rank).
    `root`    the rank the output is gathered to (None = every rank): a rank that is not the root then needs only ITS OWN
              range of the output in HBM (`alloc_out`).
    `world` / `rank` given explicitly: one rank's share WITHOUT a communicator (hsrans_sharded_create_rank; decode only) —
              tests run every rank's GPU side on one GPU that way."""

    uses_c_abi = True

    def __init__(self, ctx: "api.Context", plan, group=None, parts: int = 1, weights=None, root: int | None = None, world: int | None = None,
                 rank: int | None = None):
        self.ctx, self.group, self.root = ctx, group, root
        explicit = world is not None
        self.world = world if explicit else dist.get_world_size(group)
        self.rank = rank if explicit else dist.get_rank(group)
        self.comm = None if explicit else comm_for(ctx, group)
        self.c = api.Sharded(ctx, plan, max(1, parts), weights, root, comm=self.comm, rank=self.rank, world=self.world)
        self.layout = ShardLayout(plan, self.world, parts, weights)  # (the same C arithmetic, as Python lists)
        assert self.layout.sub_runs == [[(f, c) for f, c, _, _ in subs] for subs in self.c.shards]
        self.runs, self.ranges = self.layout.runs, self.layout.ranges
        info = self.c.info
        self.total, self.stream_len = int(info["decoded_length"]), int(info["stream_length"])
        self.first, self.count = self.runs[self.rank]
        self.window = (int(info["window_begin"]), int(info["window_end"]))
        self.out_base, self.out_len = int(info["out_base"]), int(info["out_length"])
        self.part_dplans = [self.c.part_plan(k) for k in range(self.layout.parts)]
        self._plan = plan
        self._dplan = None

    @property
    def dplan(self):
        """Device plan of this rank's whole chain run (None for a rank without chains), made on first use (decode_window)."""
        if self._dplan is None and self.count:
            self._dplan = self.ctx.make_device_plan(api.plan_slice(self._plan, self.first, self.count))
        return self._dplan

    def launch_info(self) -> dict | None:
        """Kernel geometry of what step() launches on this rank: the first sub-run's device plan (None: a rank without chains)."""
        dp = self.c.whole_plan() or next((p for p in self.part_dplans if p is not None), None)
        return dp.launch_info() if dp is not None else None

    def alloc_out(self, device) -> torch.Tensor:
        return torch.zeros(max(self.out_len, 4), dtype=torch.uint8, device=device)

    def upload_window(self, host_stream, device, side_stream: "torch.cuda.Stream | None" = None) -> torch.Tensor:
        """Device copy of this rank's window of the stream (hsrans_sharded_info: window_begin .. window_end) — the buffer is as
        long as the window, not as the stream.  The copy goes through `side_stream` when given."""
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

    def step(self, d_window: torch.Tensor, out: torch.Tensor, gather: bool = True) -> torch.Tensor:
        """One decode of the stream (hsrans_decode_sharded): this rank's sub-runs, their ranges exchanged (to `root`, or to everyone)
        behind the decode of the next sub-run.  `out` = alloc_out()'s buffer."""
        assert out.numel() >= self.out_len
        self.c.decode(d_window, out, api.SHARD_DECODE_AND_EXCHANGE if gather else api.SHARD_DECODE_ONLY)
        return out

    def exchange(self, d_window: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
        """The exchange alone, of ranges an earlier step(gather=False) decoded (bench.py times the two legs apart)."""
        self.c.decode(d_window, out, api.SHARD_EXCHANGE_ONLY)
        return out

    # -- the stream is already in this rank's HBM, whole --------------------------------------------------------------
    def decode(self, d_stream: torch.Tensor, out: torch.Tensor, gather: bool = True) -> torch.Tensor:
        lo, _hi = self.window
        return self.step(d_stream[lo:], out, gather)  # (window_begin is a multiple of 16: the slice keeps the alignment)

    def decode_window(self, d_window: torch.Tensor, out: torch.Tensor, gather: bool = True) -> torch.Tensor:
        return self.step(d_window, out, gather)

    def status_tensor(self, device) -> torch.Tensor:
        """This rank's device status (hsrans_sharded_status) as one int32 tensor on `device`."""
        st = 1 if self.c.status() else 0
        if self._dplan is not None:
            st |= 1 if self.ctx.status(self._dplan) else 0
        return torch.tensor([st], dtype=torch.int32, device=device)

    def global_status(self) -> int:
        """The launches' device status OR-ed over all ranks (one small all-reduce): every rank learns whether any rank's kernel
        met a malformed histogram / block header.  Synchronises; call it once after a batch of steps, not per step."""
        if self.comm is None:
            return int(self.status_tensor(torch.device("cpu")).item())
        dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(self.group) == "nccl" else torch.device("cpu")
        t = self.status_tensor(dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        return int(t.item())

    def check(self) -> None:
        if self.c.status() != 0 or (self._dplan is not None and self.ctx.status(self._dplan) != 0):
            raise api.HsransError("device reported a malformed histogram / block header")


class HostRehearsalDecoder:
    """The same layout, sub-runs and pipelined exchange with NO GPU: every sub-run is decoded by the library's host SIMD
    decoder (hsrans_decode_cpu on the sub-run's plan slice) into CPU tensors, the exchange runs over whatever backend the
    process group has (gloo).  For rehearsing the multi-rank logic where there is no GPU — tests/ and `bench.py --rehearse` —
    and never selected by anything on its own: the product's decode entries are the GPU ones."""

    def __init__(self, plan, container: int, states: int, bits: int, group=None, parts: int = 1, weights=None, root: int | None = None):
        self.group, self.root = group, root
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.codec = (container, states, bits)
        self.layout = ShardLayout(plan, self.world, parts, weights)
        self.runs, self.ranges, self.total = self.layout.runs, self.layout.ranges, self.layout.total
        self.first, self.count = self.runs[self.rank]
        self.window = self.layout.windows[self.rank]
        self.part_plans = [api.plan_slice(plan, f, c) if c else None for f, c in self.layout.sub_runs[self.rank]]
        self.out_base, self.out_len = 0, self.total
        if root is not None and self.rank != root:
            b, e = self.ranges[self.rank]
            self.out_base, self.out_len = b, e - b
        self._scratch = np.zeros(self.total + 64, np.uint8)

    def alloc_out(self, device=None) -> torch.Tensor:
        return torch.zeros(max(self.out_len, 4), dtype=torch.uint8)

    def upload_window(self, host_stream, device=None, side_stream=None) -> np.ndarray:
        """The rank's 'upload': a copy of the stream that is junk outside the bytes its chains may read."""
        stream = api._u8(host_stream)
        masked = np.full_like(stream, 0xEE)
        lo, hi = self.window
        masked[lo:hi] = stream[lo:hi]
        return masked

    def launch_part(self, k: int, window: np.ndarray, out: torch.Tensor) -> None:
        plan = self.part_plans[k]
        if plan is None:
            return
        container, states, bits = self.codec
        r = api.load_library().hsrans_decode_cpu(-1, 1, container, states, bits, api._p(window), window.size, api._p(self._scratch), self.total, api._p(plan), plan.size)
        if r != self.total:
            raise api.HsransError("hsrans_decode_cpu failed on a plan slice")
        b, e = self.layout.sub_ranges[self.rank][k]
        out[b - self.out_base:e - self.out_base] = torch.from_numpy(self._scratch[b:e])

    def step(self, window: np.ndarray, out: torch.Tensor, gather: bool = True) -> torch.Tensor:
        if not gather:
            for k in range(self.layout.parts):
                self.launch_part(k, window, out)
            return out
        return pipelined_gather(out, self.layout, lambda k: self.launch_part(k, window, out), self.group, self.root, self.out_base)

    def global_status(self) -> int:
        t = torch.zeros(1, dtype=torch.int32)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        return int(t.item())


def decode_sharded(ctx: "api.Context", d_stream: torch.Tensor, stream_length: int, plan, gather: bool = True, group=None) -> torch.Tensor:
    """Every rank holds the compressed stream in HBM and the (host) plan; rank r decodes chain run r.
    Returns the full decoded tensor (gather=True) or the local buffer with only this rank's range filled.
    One-shot convenience over ShardedDecoder (which keeps the device plan for repeated decodes)."""
    dec = ShardedDecoder(ctx, plan, group)
    out = torch.empty(dec.total, dtype=torch.uint8, device=d_stream.device)
    dec.decode(d_stream, out, gather=gather)
    torch.cuda.synchronize()
    dec.check()
    return out


def decode_sharded_from_host(ctx: "api.Context", host_stream, plan, gather: bool = True, group=None, parts: int = 1, root: int | None = None,
                             weights=None) -> torch.Tensor:
    """BASELINE config 4/5 shape: the stream lives in host memory on every rank, each rank uploads only the window its chains
    read (on a side stream), decodes its chains, and the decoded ranges are exchanged (pipelined over `parts` sub-runs).
    Returns the rank's output buffer: the whole output, or — a root gather on a rank that is not the root — its own range."""
    dec = ShardedDecoder(ctx, plan, group, parts=parts, weights=weights, root=root)
    dev = torch.device("cuda", torch.cuda.current_device())
    side = torch.cuda.Stream(device=dev)
    d_window = dec.upload_window(host_stream, dev, side)
    out = dec.alloc_out(dev)
    torch.cuda.current_stream(dev).wait_stream(side)
    dec.step(d_window, out, gather=gather)
    dec.check()
    return out
