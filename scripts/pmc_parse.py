"""Summarise rocprofv3 --pmc output: per SpMV dispatch (in launch order) the counter value and duration.
  usage: python scripts/pmc_parse.py <dir> <counter>"""
import csv
import glob
import sys

d, counter = sys.argv[1], sys.argv[2]     # counter: a name, or ALL for every counter in the pass
rows = []
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            if (counter == "ALL" or r.get("Counter_Name") == counter) and "spmv" in r.get("Kernel_Name", ""):
                rows.append((int(r["Dispatch_Id"]), r["Kernel_Name"][:44] + " " + r["Counter_Name"], float(r["Counter_Value"])))
dur = {}
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            dur[int(r["Dispatch_Id"])] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
for did, name, val in sorted(rows):
    print("%6d %-72s %.4g  %.0f us" % (did, name, val, dur.get(did, -1)))
