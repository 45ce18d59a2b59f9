import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import hypersonic_rans_amd as H
from hypersonic_rans_amd import synth, api
ctx = H.Context(0)
zipf = synth.enwik8_shaped(3_000_000, seed=5)
for states in (32, 64):
    for bits, n, interval in ((11, 3_000_000, 32), (14, 1 << 20, 64), (11, 65560, 4)):
        d = zipf[:n]
        s = H.encode(H.MT, states, bits, d)
        d_in = torch.from_numpy(np.concatenate([s, np.zeros((-s.size) % 16, np.uint8)])).cuda()
        base = ctx.make_device_plan_from_stream(H.MT, states, bits, d_in, s.size, n)
        out = torch.zeros(n, dtype=torch.uint8, device="cuda")
        indexed = ctx.decode_device_indexing(base, d_in, out, interval, stream_length=s.size)
        want = ctx.index_build(H.MT, states, bits, s, interval)
        got = ctx.read_device_plan(indexed, capacity=want.size + 4096)
        print(states, bits, n, interval, "sizes", got.size, want.size, "equal", np.array_equal(got, want))
        if got.size == want.size and not np.array_equal(got, want):
            diff = np.nonzero(got != want)[0]
            print("  ndiff", diff.size, "first", diff[:12], "last", diff[-3:])
            hw = api.plan_tables(want); hg = api.plan_tables(got)
            print("  header want", hw[0]); print("  header got ", hg[0])
            for name, a, b in (("chains", hw[1], hg[1]), ("pieces", hw[2], hg[2])):
                bad = [i for i in range(len(a)) if a[i] != b[i]]
                print("  ", name, "bad", len(bad), bad[:5])
                for i in bad[:3]:
                    print("     want", a[i]); print("     got ", b[i])
