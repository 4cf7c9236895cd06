# Rehearsal of the N=4 bench leg on a ONE-GPU box (mock RCCL transport, gloo bootstrap): functional check only.
set -e
cd $GRAFT_REPO_ROOT
g++ -O2 -fPIC -shared -std=c++17 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include tests/mock_rccl/mock_rccl.cpp -o /tmp/libmock_rccl.so -L/opt/rocm/lib -lamdhip64 -lrt -lpthread
export SPRS_RCCL_LIB=/tmp/libmock_rccl.so SPRS_BENCH_DEVICE=0 OMP_NUM_THREADS=1
for ex in halo allgather; do
timeout -k 10 400 python bench.py --gpus 4 --steps 20 --warmup 3 --grid 200x200x64 --dist-backend gloo --exchange $ex --no-cpu-baseline > gpurun_out/rehearse4_$ex.json 2> gpurun_out/rehearse4_$ex.err || { tail -20 gpurun_out/rehearse4_$ex.err; exit 1; }
python scripts/show_bench.py gpurun_out/rehearse4_$ex.json
done
