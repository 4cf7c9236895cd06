# Profiles of record (rounds 2, 3) (run on the GPU box): kernel-trace stats + FETCH/WRITE PMC passes per workload.
#   usage: bash scripts/profile_all.sh   -> gpurun_out/{kernel_stats_*.csv, pmc_summary_*.json, prof_*_stats.json}
set -e
bash scripts/profile_bench.sh cfg5_pair
bash scripts/profile_bench.sh cfg5_csr --stream csr
bash scripts/profile_bench.sh cfg5_random --values random
bash scripts/profile_bench.sh cfg3_banded --workload banded --steps 500 --warmup 50
bash scripts/profile_bench.sh cfg4_complex --workload complex --steps 500 --warmup 50
bash scripts/profile_bench.sh cfg2_poisson2d --workload poisson2d --steps 500 --warmup 50
