import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np, torch
import hypersonic_rans_amd as H
from hypersonic_rans_amd import synth
from test_gpu_fuzz import _mutate
use_k2 = int(sys.argv[1]) if len(sys.argv) > 1 else 1
container, states, bits = H.MT, 64, 11
ctx = H.Context(0)
rng = np.random.default_rng(1234 + container * 10 + states)
data = synth.nonstationary(200_000, seed=17)
stream = H.encode(container, states, bits, data, block_size=16384)
n = data.size
L = ctx.L
out = np.zeros(n + 4096, np.uint8)
def valid_ok():
    r = L.hsrans_decode_host(ctx.handle, container, states, bits, H.api._p(stream), stream.size, H.api._p(out), n, None, 0)
    return r == n and np.array_equal(out[:n], data)
print("before:", valid_ok())
for it in range(120):
    s = _mutate(rng, stream)
    r = L.hsrans_decode_host(ctx.handle, container, states, bits, H.api._p(s), s.size, H.api._p(out), n, None, 0)
    a = valid_ok()
    k2 = None
    if use_k2 and it % 4 == 0:
        d = torch.from_numpy(np.concatenate([s, np.zeros((-s.size) % 16 + 16, np.uint8)])).cuda()
        try:
            dp = ctx.make_device_plan_from_stream(container, states, bits, d, s.size, n)
            d_out = torch.full((n + 4096,), 0xCC, dtype=torch.uint8, device="cuda")
            ctx.decode_device(dp, d, d_out[:n], stream_length=s.size)
            k2 = ctx.status(dp)
        except H.HsransError as e:
            k2 = "err " + str(e)[-30:]
    b = valid_ok()
    if not a or not b:
        print("iteration", it, "mutated r =", r, "valid after host decode:", a, "k2:", k2, "valid after k2:", b, "diff bytes", int((s[:min(s.size, stream.size)] != stream[:min(s.size, stream.size)]).sum()), "len", s.size, stream.size)
        # one more try, and a fresh context
        print("  retry:", valid_ok(), " fresh ctx:", (lambda c2: c2.L.hsrans_decode_host(c2.handle, container, states, bits, H.api._p(stream), stream.size, H.api._p(out), n, None, 0))(H.Context(0)))
        break
else:
    print("all 120 iterations fine")
