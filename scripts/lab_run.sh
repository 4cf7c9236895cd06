# usage (on the GPU box): bash scripts/lab_run.sh <tag> [filter] [pmc-filter]
# runs scripts/spmv_lab (timing), then one rocprofv3 --pmc FETCH_SIZE pass over the variants matching pmc-filter
tag=$1; filter=${2:-}; pf=${3:-/1024}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 240 scripts/spmv_lab "$filter" 12 > gpurun_out/lab_$tag.txt 2>&1 || { tail -5 gpurun_out/lab_$tag.txt; exit 1; }
cat gpurun_out/lab_$tag.txt
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/labpmc_$tag -- scripts/spmv_lab "$pf" 1 > gpurun_out/labpmc_$tag.log 2>&1 || { tail -5 gpurun_out/labpmc_$tag.log; exit 1; }
python3 scripts/lab_pmc.py gpurun_out/labpmc_$tag gpurun_out/labpmc_$tag.log | tee gpurun_out/labpmc_$tag.txt
