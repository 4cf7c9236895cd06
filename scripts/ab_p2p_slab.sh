# The per-rank problem of N = 8 on one GPU (500x500x25 slab): plain path, distributed path with ncclAllReduce hand-offs
# (world 1: RCCL short-circuits), distributed path with the peer-to-peer mailboxes (self-mailbox: post + poll cost)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for v in "plain:" "dist_rccl:--force-dist --set p2p_allreduce=0" "dist_p2p:--force-dist"; do
  name=${v%%:*}; flags=${v#*:}
  timeout -k 10 200 python bench.py --grid 500x500x25 --steps 200 --warmup 20 --no-cpu-baseline --no-also $flags > gpurun_out/ab_p2p_$name.json 2>/dev/null || exit 1
  python - <<PY
import json
d=json.load(open("gpurun_out/ab_p2p_$name.json"))
print("$name: %.1f it/s  %.1f us/iteration  %s" % (d["value"], d["ms_per_step"]*1e3, d.get("scalar_handoff","")[:40]))
PY
done; done
