"""Device-memory hygiene of the C ABI: repeated create / destroy cycles must not leak HBM."""
import numpy as np
import pytest

import hypersonic_rans_amd as H
from hypersonic_rans_amd import synth

pytestmark = pytest.mark.gpu


def test_no_device_memory_leak_over_create_destroy_cycles(gpu_ctx):
    import torch

    d = synth.enwik8_shaped(2_000_000, seed=9)
    s, plan = H.encode(H.RAW, 64, 11, d, index_interval=32)
    sm, planm = H.encode(H.MT, 64, 11, d, index_interval=32, block_size=65536)
    d_in = torch.from_numpy(d).cuda()
    d_enc = torch.empty(H.capacity(H.MT, 64, d.size), dtype=torch.uint8, device="cuda")
    d_out = torch.zeros(d.size, dtype=torch.uint8, device="cuda")

    def cycle():
        for p in (plan, planm):
            dp = gpu_ctx.make_device_plan(p)
            dp.close()
        m, dp = gpu_ctx.encode_device(H.MT, 64, 11, d_in, d_enc, block_size=65536, index_interval=32, want_plan=True)
        gpu_ctx.decode_device(dp, d_enc, d_out, stream_length=m)
        assert gpu_ctx.status(dp) == 0
        dp.close()
        dp = gpu_ctx.make_device_plan_from_stream(H.MT, 64, 11, d_enc, m, d.size)
        dp.close()
        gpu_ctx.index_build(H.RAW, 64, 11, s, 64)
        r, _ = gpu_ctx.decode_host(H.RAW, 64, 11, s, d.size, plan=plan)
        assert r == d.size

    for _ in range(3):  # warm the context's own staging buffers
        cycle()
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    for _ in range(40):
        cycle()
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert free0 - free1 < 8 << 20, (free0, free1)
    assert torch.equal(d_out, d_in)
