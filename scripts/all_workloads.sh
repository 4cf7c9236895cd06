set -e
run() { echo "== $*"; python bench.py "$@" --no-cpu-baseline > gpurun_out/aw.json 2> gpurun_out/aw.err || { tail -5 gpurun_out/aw.err; exit 1; }; python scripts/show_bench.py gpurun_out/aw.json | head -3; }
run --workload bench100
run --workload poisson2d --steps 200 --warmup 20
run --workload banded --steps 200 --warmup 20
run --workload complex --steps 200 --warmup 20
run --workload poisson3d --stream csr --steps 10 --warmup 2 --no-also
run --workload poisson3d --stream offsets --steps 10 --warmup 2 --no-also
run --workload poisson3d --values random --steps 10 --warmup 2 --no-also
run --workload poisson3d --grid 500x500x25 --force-dist --steps 50 --warmup 5 --no-also
run --workload poisson3d --grid 300x300x100 --steps 20 --warmup 5 --no-also
