for n in 1000 96 720 600 1200 1536 2000 250 360 1800 1920 500 120 1280; do
for r in 2 3 4 5; do for u in 1 2; do
FXC_RTC_R0=$r FXC_RTC_U=$u python tools/bench_spec.py --child --dev --reps 3 --cases $n 2>&1 | grep -v amdgpu.ids | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('N', d['nchan'], 'r0', $r, 'u', $u, 'vgprs', d['vgprs'], 'block', d['block'], 'lds', d['lds'], 'ms', d['median_ms'], d['frac_of_8TBs'])
"
done; done; done
