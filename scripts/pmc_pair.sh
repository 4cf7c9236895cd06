# usage: bash scripts/pmc_pair.sh <tag> [knobs]   -> gpurun_out/pmcpair_<tag>.txt : per-counter mean over the spmv_pair2 launches
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
rm -f gpurun_out/pmcpair_$tag.txt
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" \
           "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM" \
           "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES SQ_INST_CYCLES_VMEM_RD SQ_BUSY_CU_CYCLES" \
           "TA_BUSY_avr TA_TA_BUSY_sum TA_TOTAL_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum" \
           "TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCR_TCP_STALL_CYCLES_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum TCC_TAG_STALL_sum TCC_BUSY_avr" \
           "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_CYCLES"; do
  i=$((i+1))
  timeout -k 10 120 rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/pmcpair_${tag}_$i -- python3 scripts/pmc_pair.py 12 "$@" > gpurun_out/pmcpair_${tag}_$i.log 2>&1 || { echo "pass $i failed"; tail -3 gpurun_out/pmcpair_${tag}_$i.log; continue; }
  python3 - <<PY >> gpurun_out/pmcpair_$tag.txt
import csv, glob, collections, statistics
vals = collections.defaultdict(list); dur = []
for f in glob.glob("gpurun_out/pmcpair_${tag}_$i/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "spmv_pair2" in r["Kernel_Name"]: vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob("gpurun_out/pmcpair_${tag}_$i/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "spmv_pair2" in r["Kernel_Name"]: dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k in sorted(vals): print("%-40s %.6g  (n=%d)" % (k, statistics.median(vals[k]), len(vals[k])))
if dur: print("%-40s %.1f us" % ("  kernel duration under this pass", statistics.median(dur)))
PY
  rm -rf gpurun_out/pmcpair_${tag}_$i
done
cat gpurun_out/pmcpair_$tag.txt
