set -e
run() { name=$1; shift; timeout -k 10 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-also --stream dict "$@" > gpurun_out/sw_$name.json 2> gpurun_out/sw_$name.err; }
run base --set spmv_grid=1536
run chunk --set spmv_grid=1536 --set xcd_chunk=1
run period --set spmv_grid=1536 --set spmv_strip=1
run strip2k --set spmv_grid=1536 --set spmv_strip=2048
run strip8k --set spmv_grid=1536 --set spmv_strip=8192
run strip32k --set spmv_grid=1536 --set spmv_strip=32768
python scripts/show_bench.py gpurun_out/sw_base.json gpurun_out/sw_chunk.json gpurun_out/sw_period.json gpurun_out/sw_strip2k.json gpurun_out/sw_strip8k.json gpurun_out/sw_strip32k.json
