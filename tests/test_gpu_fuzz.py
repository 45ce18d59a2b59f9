"""Mutated streams through the GPU entries: whatever the bytes say, a decode returns (the decoded length, or 0 / an error code) — no
fault, no hang, no write outside the output buffer (guard bytes behind it stay untouched).  The plans the kernels run come from the
host planner / validator (fuzzed under sanitizers in tests/test_fuzz_host.py); what is exercised HERE are the parsers that run on
the device: the block_ walk over inline headers (run_block_walk), the mt_ header chase (K2, hsrans_dplan_create_from_device_stream),
the in-kernel histogram checks and the stream windows' bounds."""
import numpy as np
import pytest
import torch

import hypersonic_rans_amd as H
from hypersonic_rans_amd import synth

pytestmark = pytest.mark.gpu

EDGES = (0, 1, 0x7FFF, 0x8000, 0xFFFF, 0x7FFFFFFF, 0x80000000, 0xFFFFFFFF, 0xFFFFFFFF00000000, 0x7FFFFFFFFFFFFFFF, 0xFFFFFFFFFFFFFFFF)


def _mutate(rng, stream):
    s = stream.copy()
    for _ in range(int(rng.integers(1, 4))):
        kind = int(rng.integers(0, 5))
        at = int(rng.integers(0, s.size))
        if kind == 0:
            s[at] ^= np.uint8(1 << int(rng.integers(0, 8)))
        elif kind == 1:
            w = 1 << int(rng.integers(1, 4))
            a = at // w * w
            if a + w <= s.size:
                s[a:a + w] = np.frombuffer(np.uint64(EDGES[int(rng.integers(0, len(EDGES)))]).tobytes(), np.uint8)[:w]
        elif kind == 2:
            s = s[:max(at, 1)].copy()
        elif kind == 3:
            a = at // 8 * 8
            if a + 8 <= s.size:
                v = int(s[a:a + 8].view(np.uint64)[0])
                s[a:a + 8] = np.frombuffer(np.uint64((v + (64 if rng.integers(0, 2) else -64)) % (1 << 64)).tobytes(), np.uint8)
        else:
            n = int(rng.integers(1, 64))
            s[at:at + n] = rng.integers(0, 256, size=s[at:at + n].size, dtype=np.uint8)
    return s


@pytest.mark.parametrize("container,states,bits", ((H.BLOCK, 64, 11), (H.MT, 64, 11), (H.RAW, 64, 12), (H.BLOCK, 32, 13), (H.MT, 32, 10)))
def test_mutated_streams_never_fault_or_write_outside(gpu_ctx, container, states, bits):
    rng = np.random.default_rng(1234 + container * 10 + states)
    data = synth.nonstationary(200_000, seed=17)
    stream = H.encode(container, states, bits, data, block_size=0 if container == H.RAW else 16384)
    n = data.size
    L = gpu_ctx.L
    out = np.zeros(n + 4096, np.uint8)
    ok = bad = 0
    for it in range(120):
        s = _mutate(rng, stream)
        out[:] = 0xCC
        r = L.hsrans_decode_host(gpu_ctx.handle, container, states, bits, H.api._p(s), s.size, H.api._p(out), n, None, 0)
        assert r in (0, n)
        assert (out[n:] == 0xCC).all(), "a decode wrote behind its output buffer"
        ok += r == n
        bad += r == 0
        if container == H.MT and it % 4 == 0:  # the device-side header chase on the same bytes
            d = torch.from_numpy(np.concatenate([s, np.zeros((-s.size) % 16 + 16, np.uint8)])).cuda()
            try:
                dp = gpu_ctx.make_device_plan_from_stream(container, states, bits, d, s.size, n)
                d_out = torch.full((n + 4096,), 0xCC, dtype=torch.uint8, device="cuda")
                gpu_ctx.decode_device(dp, d, d_out[:n], stream_length=s.size)
                gpu_ctx.status(dp)
                assert bool((d_out[n:] == 0xCC).all())
            except H.HsransError:
                pass
    torch.cuda.synchronize()
    assert ok + bad == 120
    # the unmodified stream still decodes after all that
    r = L.hsrans_decode_host(gpu_ctx.handle, container, states, bits, H.api._p(stream), stream.size, H.api._p(out), n, None, 0)
    assert r == n and np.array_equal(out[:n], data)


@pytest.mark.parametrize("container,states,bits", ((H.MT, 64, 11), (H.RAW, 64, 11), (H.MT, 32, 12)))
def test_mutated_streams_through_the_recording_first_decode_and_the_index_cache(gpu_ctx, container, states, bits):
    """Streams of >= 1 MiB take the plan-less host entry through the pass that RECORDS checkpoints and — mt_ — through the plan
    assembled on the device from whatever that pass wrote; the next call may find the cached index.  Mutated bytes, twice each,
    in ONE buffer (the cache is keyed by its address): no fault, no hang, nothing behind the output, the same result both times,
    and the unmodified stream decodes afterwards.  The device-resident form (K2 walk + hsrans_decode_device_indexing + a decode
    with the plan it left) gets every fourth mutation as well."""
    rng = np.random.default_rng(4321 + container * 10 + states)
    n = 1_500_000
    data = synth.nonstationary(n, seed=23)
    stream = H.encode(container, states, bits, data, block_size=0 if container == H.RAW else 32768)
    ctx = H.Context(0)
    L = ctx.L
    buf = np.zeros(stream.size, np.uint8)
    out = np.zeros(n + 4096, np.uint8)
    decoded = 0
    for it in range(40):
        s = _mutate(rng, stream)
        ln = s.size
        buf[:ln] = s
        results = []
        for _ in range(2):
            out[:] = 0xCC
            r = L.hsrans_decode_host(ctx.handle, container, states, bits, H.api._p(buf), ln, H.api._p(out), n, None, 0)
            assert r in (0, n) and (out[n:] == 0xCC).all(), "a decode wrote behind its output buffer"
            results.append((r, out[:n].copy() if r == n else None))
        assert results[0][0] == results[1][0] and (results[0][0] == 0 or np.array_equal(results[0][1], results[1][1])), "a cached index changed the result"
        decoded += results[0][0] == n
        if container == H.MT and it % 4 == 0:
            d = torch.from_numpy(np.concatenate([s, np.zeros((-s.size) % 16 + 16, np.uint8)])).cuda()
            try:
                base = ctx.make_device_plan_from_stream(container, states, bits, d, s.size, n)
                d_out = torch.full((n + 4096,), 0xCC, dtype=torch.uint8, device="cuda")
                indexed = ctx.decode_device_indexing(base, d, d_out[:n], 32, stream_length=s.size)
                ctx.decode_device(indexed, d, d_out[:n], stream_length=s.size)
                ctx.status(indexed)
                assert bool((d_out[n:] == 0xCC).all())
            except H.HsransError:
                pass
    torch.cuda.synchronize()
    assert decoded >= 1
    buf[:stream.size] = stream
    r = L.hsrans_decode_host(ctx.handle, container, states, bits, H.api._p(buf), stream.size, H.api._p(out), n, None, 0)
    assert r == n and np.array_equal(out[:n], data)


@pytest.mark.parametrize("form", ("one chain per wave (raw members)", "grouped (mt_ members)", "grouped (block_ members)"))
def test_mutated_member_of_a_batch_never_disturbs_the_others(gpu_ctx, form):
    """One launch, several member streams (hsrans_decode_device_batch): the bytes of ONE member are mutated (the plans stay the honest
    ones: what changes under the kernel are histograms, headers and words), 60 times.  No fault, no hang, nothing written behind any
    member's output, and every OTHER member decodes bit-exactly with a clean status word every time — a member is judged alone."""
    rng = np.random.default_rng(777 + len(form))
    container = H.RAW if "raw" in form else H.MT if "mt_" in form else H.BLOCK
    ms = []
    for i in range(4):
        n = 400_000 + 150_001 * i
        data = synth.nonstationary(n, seed=40 + i)
        if container == H.RAW:
            stream, plan = H.encode(H.RAW, 64, 11, data, index_interval=(16, 32)[i % 2])
        else:
            stream, plan = H.encode(container, 64, 11, data, block_size=(32768, 65536)[i % 2], index_interval=32)
        ms.append({"n": n, "data": data, "stream": stream, "dplan": gpu_ctx.make_device_plan(plan),
                   "d_in": torch.from_numpy(np.concatenate([stream, np.zeros((-stream.size) % 16 + 64, np.uint8)])).cuda(),
                   "d_out": torch.full((n + 4096,), 0xCC, dtype=torch.uint8, device="cuda")})
    batch = gpu_ctx.make_batch([m["dplan"] for m in ms])
    info = batch.info()
    assert info["launches"] == 1 and (info["direct_members"] == 4 if container == H.RAW else info["grouped_members"] == 4)
    lens = [m["stream"].size for m in ms]
    flagged = 0
    for it in range(60):
        victim = int(rng.integers(0, 4))
        s = _mutate(rng, ms[victim]["stream"])
        if s.size < ms[victim]["stream"].size:  # (a truncated member: the batch entry checks lengths before it launches; keep the length, junk the tail)
            s = np.concatenate([s, rng.integers(0, 256, size=ms[victim]["stream"].size - s.size, dtype=np.uint8)])
        bad = torch.from_numpy(np.concatenate([s, np.zeros((-s.size) % 16 + 64, np.uint8)])).cuda()
        for m in ms:
            m["d_out"].fill_(0xCC)
        gpu_ctx.decode_device_batch(batch, [bad if k == victim else m["d_in"] for k, m in enumerate(ms)], [m["d_out"][:m["n"]] for m in ms], stream_lengths=lens)
        torch.cuda.synchronize()
        codes = gpu_ctx.batch_status(batch)
        flagged += codes[victim] != 0
        for k, m in enumerate(ms):
            assert bool((m["d_out"][m["n"]:] == 0xCC).all()), f"member {k} wrote behind its output (iteration {it}, victim {victim})"
            if k != victim:
                assert codes[k] == 0, (it, victim, codes)
                assert np.array_equal(m["d_out"][:m["n"]].cpu().numpy(), m["data"]), f"member {k} was disturbed by member {victim}'s bytes (iteration {it})"
    # the unmodified members still decode after all that
    gpu_ctx.decode_device_batch(batch, [m["d_in"] for m in ms], [m["d_out"][:m["n"]] for m in ms], stream_lengths=lens)
    torch.cuda.synchronize()
    assert gpu_ctx.batch_status(batch) == [0] * 4
    for m in ms:
        assert np.array_equal(m["d_out"][:m["n"]].cpu().numpy(), m["data"])
