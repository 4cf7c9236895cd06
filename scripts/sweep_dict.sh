# in-solve sweep of the placement / schedule knobs for the compressed-stream SpMV (cfg 5)
set -e
run() { name=$1; shift; timeout -k 10 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-also "$@" > gpurun_out/sw_$name.json 2> gpurun_out/sw_$name.err; python scripts/show_bench.py gpurun_out/sw_$name.json | head -1; }
run base
run chunk --set xcd_chunk=1
run period --set spmv_strip=1
run strip2k --set spmv_strip=2048
run strip8k --set spmv_strip=8192
run strip32k --set spmv_strip=32768
run base2
