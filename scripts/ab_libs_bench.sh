# Same-box A/B of library builds (scripts/ab/lib_*.so, git-ignored) on the default bench command: in-solve SpMV mean and it/s.
#   usage (GPU box): bash scripts/ab_libs_bench.sh [rounds] [bench.py flags ...]
ROUNDS=${1:-2}; shift
cp sprsolve_amd/libsprsolve_hip.so /tmp/lib_keep.so
for r in $(seq 1 $ROUNDS); do
  for lib in scripts/ab/lib_*.so; do
    cp $lib sprsolve_amd/libsprsolve_hip.so
    timeout -k 10 300 python bench.py --steps 100 --no-cpu-baseline --no-also "$@" > /tmp/ab.json 2>/dev/null
    python3 -c "
import json
d=json.loads(open('/tmp/ab.json').read().strip().splitlines()[-1]); r=d['roofline']
print('$(basename $lib .so)', round(d['value'],1), 'it/s  spmv', round(r['avg_launch_us'],1), 'us')"
  done
done
cp /tmp/lib_keep.so sprsolve_amd/libsprsolve_hip.so
