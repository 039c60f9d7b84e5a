cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp CBH_BENCH_NO_DIST=1; mkdir -p gpurun_out/r06m
B="--steps 10 --warmup 2 --no-cpu-baseline --no-side-legs --serial-leg --check-first"
for rep in 1 2; do for m in default 1; do for w in C3 C2; do
  if [ $m = 1 ]; then export CBH_FLAT_MASKS=1; else unset CBH_FLAT_MASKS; fi
  timeout -k 5 200 python bench.py --workload $w $B > gpurun_out/r06m/b_${w}_$m_$rep.json 2> gpurun_out/r06m/b_${w}_$m_$rep.err
  python -c "
import json
d=json.loads([l for l in open('gpurun_out/r06m/b_${w}_$m_$rep.json') if l.startswith('{')][-1]); r=d['roofline']; s=r.get('serial') or {}
print('$w masks=$m #$rep', '%.4g dec/s' % d['value'], r['kernel'], 'frac %.3f' % r['frac'], 'by itself %.1f us' % (s.get('kernel_ms', 0) * 1e3), '|', d.get('first_batch_checked'))"
done; done; done 2>&1 | tee gpurun_out/r06m/masks.txt
