#!/usr/bin/env python3
"""k_decode_single diagnostics: how often does the decoding wave find the producer's ring short?  (HSRANS_DEBUG_STAMPS=1)"""
import ctypes, os, sys
os.environ["HSRANS_DEBUG_STAMPS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import hypersonic_rans_amd as H
from hypersonic_rans_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
bits = int(sys.argv[2]) if len(sys.argv) > 2 else 11
d = synth.enwik8_shaped(n)
s = H.encode(H.RAW, 64, bits, d)
ctx = H.Context(0)
dp = ctx.make_device_plan(H.plan_build(H.RAW, 64, bits, s))
d_in = torch.from_numpy(np.concatenate([s, np.zeros((-s.size) % 16, np.uint8)])).cuda()
d_out = torch.zeros(n, dtype=torch.uint8, device="cuda")
for _ in range(2):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); ctx.decode_device(dp, d_in, d_out, stream_length=s.size); b.record(); torch.cuda.synchronize()
assert torch.equal(d_out.cpu(), torch.from_numpy(d))
L = H.load_library()
L.hsrans_debug_read_stamps.restype = ctypes.c_size_t
L.hsrans_debug_read_stamps.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
buf = np.zeros(8, np.uint64)
L.hsrans_debug_read_stamps(dp.handle, buf.ctypes.data, 8)
ms = a.elapsed_time(b)
print(f"bits {bits}: {n / 2**20 / (ms * 1e-3):.0f} MiB/s, {ms:.2f} ms; in-kernel {(int(buf[3]) - int(buf[0])) / 100:.0f} us for {int(buf[4])} groups = {(int(buf[3]) - int(buf[0])) * 10 / max(1, int(buf[4])):.1f} ns per group; "
      f"shader clocks {int(buf[5])} = {int(buf[5]) / max(1, int(buf[4])):.0f} per group, {int(buf[5]) / max(1, (int(buf[3]) - int(buf[0])) * 10):.2f} GHz; ring count reads {int(buf[1])}, of which found the ring short {int(buf[2])}; launch {dp.launch_info()}")
