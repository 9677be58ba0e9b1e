for r in 2 3 2 3; do python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-gather --no-other-configs --ranges $r 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); p=d['pipelined_step']; print('ranges', $r, 'pipe %.1f us %.3f' % (p['ms_per_step']*1e3, p['roofline_frac']), 'single %.1f' % (d['single_stream']['ms_per_step']*1e3), flush=True)"; done
