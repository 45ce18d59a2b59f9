"""Host-resident streams: upload, decode and download overlapped on separate HIP streams (BASELINE config 5 shape).

The reference decodes host buffers in place (mt_rANS32x64_16w_decode.cpp:137-265: a thread pool over the blocks).  With the
stream and the output in (pinned) host memory the GPU version is PCIe-bound, so the three legs are pipelined over slices of
the plan's chains: slice k's compressed bytes go up on one stream while slice k-1 decodes on another and slice k-2's decoded
bytes come down on a third — the two DMA directions of the link run concurrently and the kernels hide behind them.
"""
from __future__ import annotations

import numpy as np
import torch

from . import api
from .sharded import local_range, shard_chains


class PipelinedHostDecoder:
    """Prepared once per (plan, slicing); `decode` can then be called repeatedly for streams that share the plan's layout
    (in practice: the same stream).  `host_stream` and `host_out` should be pinned (torch `pin_memory()`): with pageable
    memory the runtime stages the copies itself and the legs no longer overlap."""

    def __init__(self, ctx: "api.Context", plan, n_slices: int = 8, device: "torch.device | None" = None):
        self.ctx = ctx
        self.device = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.total = api.plan_decoded_length(plan)
        hdr, _, _ = api.plan_tables(plan)
        self.stream_len = int(hdr["stream_len"])
        runs = [(f, c) for f, c in shard_chains(plan, max(1, n_slices)) if c]
        self.slices = []
        for f, c in runs:
            self.slices.append({
                "in": [(lo, hi) for lo, hi in api.plan_stream_ranges(plan, f, c) if hi > lo],
                "out": local_range(plan, f, c),
                "dplan": ctx.make_device_plan(api.plan_slice(plan, f, c)),
            })
        self.d_stream = torch.empty(self.stream_len + (-self.stream_len) % 16, dtype=torch.uint8, device=self.device)
        self.d_out = torch.empty(self.total, dtype=torch.uint8, device=self.device)
        self.up = torch.cuda.Stream(device=self.device)
        self.down = torch.cuda.Stream(device=self.device)
        self.uploaded_bytes = sum(hi - lo for s in self.slices for lo, hi in s["in"])

    def decode(self, host_stream: torch.Tensor, host_out: torch.Tensor) -> None:
        """Asynchronous until the final synchronisation of the download stream; `host_out[:total]` holds the decoded bytes on return."""
        assert host_stream.dtype == torch.uint8 and host_out.dtype == torch.uint8 and host_out.numel() >= self.total
        main = torch.cuda.current_stream(self.device)
        self.up.wait_stream(main)      # previous use of d_stream / d_out is finished
        self.down.wait_stream(main)
        ups = []
        with torch.cuda.stream(self.up):
            for s in self.slices:
                for lo, hi in s["in"]:
                    self.d_stream[lo:hi].copy_(host_stream[lo:hi], non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(self.up)
                ups.append(ev)
        for s, ev in zip(self.slices, ups):
            main.wait_event(ev)
            self.ctx.decode_device(s["dplan"], self.d_stream, self.d_out, stream=main, stream_length=self.stream_len)
            done = torch.cuda.Event()
            done.record(main)
            b, e = s["out"]
            with torch.cuda.stream(self.down):
                self.down.wait_event(done)
                host_out[b:e].copy_(self.d_out[b:e], non_blocking=True)
        self.down.synchronize()
        for s in self.slices:
            if self.ctx.status(s["dplan"]) != 0:
                raise api.HsransError("device reported a malformed histogram / block header")


def decode_from_host_unpipelined(ctx: "api.Context", plan, host_stream: torch.Tensor, host_out: torch.Tensor, device=None) -> None:
    """The same work without overlap (upload everything, decode, download everything): the comparison point."""
    device = device if device is not None else torch.device("cuda", torch.cuda.current_device())
    total = api.plan_decoded_length(plan)
    n = host_stream.numel()
    d_stream = torch.empty(n + (-n) % 16, dtype=torch.uint8, device=device)
    d_stream[:n].copy_(host_stream, non_blocking=True)
    d_out = torch.empty(total, dtype=torch.uint8, device=device)
    dplan = ctx.make_device_plan(plan)
    ctx.decode_device(dplan, d_stream, d_out, stream_length=n)
    host_out[:total].copy_(d_out, non_blocking=True)
    torch.cuda.synchronize(device)
    if ctx.status(dplan) != 0:
        raise api.HsransError("device reported a malformed histogram / block header")
