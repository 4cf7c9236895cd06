cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_cp
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_cp -- python3 scripts/cp_probe.py > gpurun_out/prof_cp.log 2>&1 || { tail -3 gpurun_out/prof_cp.log; exit 1; }
grep -v "^W2026" gpurun_out/prof_cp.log | tail -12
python3 - <<'PY'
import csv, glob
rows = list(csv.DictReader(open(glob.glob("gpurun_out/prof_cp/*/*kernel_trace.csv")[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# consecutive runs of the same spmv kernel
cur, acc = None, []
def flush():
    if cur and len(acc) >= 50:
        a = sorted(acc)
        print("%-60s calls %4d  median %.1f us  min %.1f" % (cur[:60], len(acc), a[len(a)//2] / 1e3, a[0] / 1e3))
for r in rows:
    nm = r["Kernel_Name"]
    if "spmv" not in nm: continue
    nm = nm[nm.index("spmv"):]
    if nm != cur:
        flush(); cur, acc = nm, []
    acc.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
flush()
PY
