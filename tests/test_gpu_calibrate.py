"""hsrans_ctx_calibrate: the per-device fit of the one-chain-per-wave index's class lengths."""
import numpy as np
import pytest

import hypersonic_rans_amd as H
from hypersonic_rans_amd import synth

pytestmark = pytest.mark.gpu


def test_calibration_keeps_decodes_bit_exact_and_does_not_finish_later():
    ctx = H.Context(0)
    data = synth.enwik8_shaped(12_000_000, seed=5)
    before = H.index_boundaries(64, 11, data.size, ctx)
    rep = ctx.calibrate(bits=11, iterations=3)
    assert len(rep["class_weights"]) == 8 and abs(sum(rep["class_weights"]) - 8000) <= 8
    assert all(100 <= w <= 3000 for w in rep["class_weights"]), rep
    # the best iteration is kept: never worse than the first one (the compiled-in lengths)
    assert rep["last_wave_us_after"] <= rep["last_wave_us_before"] + 1e-9, rep
    after = H.index_boundaries(64, 11, data.size, ctx)
    assert after.size == before.size  # one chain per resident wavefront either way
    # an index made with the calibrated lengths decodes bit-exactly, and the launch info names the lengths it was shaped with
    stream, plan = H.encode(H.RAW, 64, 11, data, index_groups=after)
    got = ctx.decode(H.RAW, 64, 11, stream, plan=plan)
    assert np.array_equal(got, data)
    import torch
    dp = ctx.make_device_plan(plan)
    d_in = torch.from_numpy(np.concatenate([stream, np.zeros((-stream.size) % 16, np.uint8)])).cuda()
    d_out = torch.zeros(data.size, dtype=torch.uint8, device="cuda")
    ctx.decode_device(dp, d_in, d_out, stream_length=stream.size)
    assert ctx.status(dp) == 0 and np.array_equal(d_out.cpu().numpy(), data)
    assert dp.launch_info()["class_weights"] == rep["class_weights"]
    # other contexts are untouched (nothing about a device is process-global)
    other = H.Context(0)
    assert np.array_equal(H.index_boundaries(64, 11, data.size, other), before)
    with pytest.raises(H.HsransError):
        ctx.calibrate(bits=14)
