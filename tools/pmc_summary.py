"""per-kernel sums of a rocprofv3 --pmc counter_collection.csv: one line per (kernel, counter), averaged over dispatches
    python tools/pmc_summary.py <dir or csv> [kernel-name-substring]"""
import csv, glob, os, sys
from collections import defaultdict
path = sys.argv[1]
files = [path] if path.endswith(".csv") else glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True)
sub = sys.argv[2] if len(sys.argv) > 2 else ""
acc = defaultdict(lambda: defaultdict(float)); nd = defaultdict(set); ndc = defaultdict(lambda: defaultdict(set))
for f in files:
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "")
        if sub not in k: continue
        k = k.split("(")[0][:60]
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"]); nd[k].add((f, row["Dispatch_Id"]))
        ndc[k][row["Counter_Name"]].add((f, row["Dispatch_Id"]))   # (a counter is in one pass only: average over ITS dispatches)
for k in sorted(acc):
    n = max(len(nd[k]), 1)
    c = {name: v / max(len(ndc[k][name]), 1) for name, v in acc[k].items()}
    print("%s  (%d dispatches)" % (k, n))
    for name in sorted(c): print("    %-24s %.6g" % (name, c[name]))
    if "SQ_ACTIVE_INST_VALU" in c and "SQ_WAVE_CYCLES" in c and c["SQ_WAVE_CYCLES"] > 0:
        # both in quad-cycles summed over waves: the share of its life a wave spends issuing VALU; times the waves that share a SIMD
        # = the share of the SIMD's issue slots in use (GRBM_GUI_ACTIVE is summed over XCDs and follows the clock: not used as a ruler)
        print("    a wave issues VALU in %.3f of its cycles" % (c["SQ_ACTIVE_INST_VALU"] / c["SQ_WAVE_CYCLES"]))
    if "SQ_WAVE_CYCLES" in c and c["SQ_WAVE_CYCLES"] > 0:
        w = c["SQ_WAVE_CYCLES"]
        print("    of all wave-cycles: issuing %.3f, parked (s_waitcnt / barrier) %.3f, issue-stalled %.3f" % (
            c.get("SQ_ACTIVE_INST_ANY", 0) / w, c.get("SQ_WAIT_ANY", 0) / w, c.get("SQ_WAIT_INST_ANY", 0) / w))
    if "SQ_LDS_BANK_CONFLICT" in c and c.get("SQ_LDS_IDX_ACTIVE", 0) > 0:
        print("    LDS: %.3f of its active cycles are bank-conflict cycles" % (c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"]))
