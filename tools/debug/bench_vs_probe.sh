for p in 1 4; do timeout 300 python bench.py --no-cpu --no-single --pairs $p 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read()); rf=r['roofline']
print('pairs',r['config']['pairs'],'value',round(r['value']),'kernel_ms_avg',rf['kernel_ms_avg'],'frac',round(rf['frac'],4),'warm',rf['warm']['kernel_ms_avg'], 'single-launch min', rf.get('kernel_ms_single_launch_min'))"; done
timeout 300 python tools/cold_cache.py --quick 2>/dev/null | grep case
