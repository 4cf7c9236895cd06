# Rehearsal of the N=2 bench leg on a ONE-GPU box: two ranks share the card, RCCL replaced by tests/mock_rccl
# (SPRS_RCCL_LIB), torch.distributed on gloo.  Functional check only — timings mean nothing.
set -e
g++ -O2 -fPIC -shared -std=c++17 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include tests/mock_rccl/mock_rccl.cpp -o /tmp/libmock_rccl.so -L/opt/rocm/lib -lamdhip64 -lrt -lpthread
export SPRS_RCCL_LIB=/tmp/libmock_rccl.so SPRS_BENCH_DEVICE=0 OMP_NUM_THREADS=1
timeout -k 10 280 python \
    bench.py --gpus 2 --steps 20 --warmup 3 --grid 200x200x64 --dist-backend gloo "$@"
