python - <<'PY'
import sys, json, os
sys.path.insert(0, '.')
import numpy as np, torch
import hypersonic_rans_amd as H
from hypersonic_rans_amd import synth
n=100_000_000
ctx=H.Context(0)
data=synth.enwik8_shaped(n)
for S,bits in ((64,11),(32,11),(64,12),(32,12),(64,13),(32,10)):
    s,plan=H.encode(H.RAW,S,bits,data,index_interval=32)
    d_in=torch.from_numpy(np.concatenate([s,np.zeros((-s.size)%16,np.uint8)])).cuda()
    d_out=torch.zeros(n,dtype=torch.uint8,device='cuda')
    dp=ctx.make_device_plan(plan)
    ctx.decode_device(dp,d_in,d_out,stream_length=s.size); torch.cuda.synchronize()
    ok=bool(np.array_equal(d_out.cpu().numpy(),data)) and ctx.status(dp)==0
    for _ in range(10): ctx.decode_device(dp,d_in,d_out,stream_length=s.size)
    a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(40): ctx.decode_device(dp,d_in,d_out,stream_length=s.size)
    b.record(); torch.cuda.synchronize()
    us=a.elapsed_time(b)/40*1e3
    print(json.dumps({"states":S,"bits":bits,"index":"G=32","bit_exact":ok,"kernel_us":round(us,2),"frac":round((s.size+n)/(us*1e-6)/8e12,4)}),flush=True)
PY
