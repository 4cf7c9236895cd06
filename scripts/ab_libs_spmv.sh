# Same-box A/B of library builds on the back-to-back cfg-5 SpMV: every lib in scripts/ab/lib_*.so (git-ignored builds) is
# copied over sprsolve_amd/libsprsolve_hip.so in turn, ROUNDS times, and scripts/time_spmv.py is run on it.
#   usage (on the GPU box): bash scripts/ab_libs_spmv.sh [rounds] [stream] [KEY=VALUE ...]
ROUNDS=${1:-3}; STREAM=${2:-pair}; shift 2 2>/dev/null
cp sprsolve_amd/libsprsolve_hip.so /tmp/lib_keep.so
for r in $(seq 1 $ROUNDS); do
  for lib in scripts/ab/lib_*.so; do
    cp $lib sprsolve_amd/libsprsolve_hip.so
    echo -n "$(basename $lib .so)  "
    timeout -k 10 300 python3 scripts/time_spmv.py $STREAM 50 "$@" 2>&1 | tail -1
  done
done
cp /tmp/lib_keep.so sprsolve_amd/libsprsolve_hip.so
