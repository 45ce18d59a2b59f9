# quick look at one histogram width: correctness (sweep_configs checks bit-exactness) and time.  bash tools/debug/bits_quick.sh 13
python - "$@" <<'PY'
import sys, json, os
sys.path.insert(0, '.')
import numpy as np, torch
import hypersonic_rans_amd as H
from hypersonic_rans_amd import synth
n=100_000_000
S=int(os.environ.get("STATES","64"))
ctx=H.Context(0)
data=synth.enwik8_shaped(n)
for bits in [int(x) for x in sys.argv[1:]] or [13]:
    g=H.index_boundaries(S,bits,n,ctx)
    s,plan=H.encode(H.RAW,S,bits,data,index_groups=g)
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
    print(json.dumps({"states":S,"bits":bits,"bit_exact":ok,"kernel_us":round(us,2),"frac":round((s.size+n)/(us*1e-6)/8e12,4),"launch":dp.launch_info()}),flush=True)
PY
