"""per-kernel sums of a rocprofv3 --pmc counter_collection.csv: one line per (kernel, counter), averaged over dispatches
    python tools/pmc_summary.py <dir or csv> [kernel-name-substring]"""
import csv, glob, os, sys
from collections import defaultdict
path = sys.argv[1]
files = [path] if path.endswith(".csv") else glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True)
sub = sys.argv[2] if len(sys.argv) > 2 else ""
acc = defaultdict(lambda: defaultdict(float)); nd = defaultdict(set)
for f in files:
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "")
        if sub not in k: continue
        k = k.split("(")[0][:60]
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"]); nd[k].add((f, row["Dispatch_Id"]))
for k in sorted(acc):
    n = max(len(nd[k]), 1)
    c = {name: v / n for name, v in acc[k].items()}
    print("%s  (%d dispatches)" % (k, n))
    for name in sorted(c): print("    %-24s %.6g" % (name, c[name]))
    if "SQ_ACTIVE_INST_VALU" in c and "GRBM_GUI_ACTIVE" in c:
        # SQ_ACTIVE_INST_* count quad-cycles summed over waves; 1024 SIMDs can each issue VALU every cycle
        print("    VALU issue utilisation   %.3f  (4 x SQ_ACTIVE_INST_VALU / (1024 SIMDs x GRBM_GUI_ACTIVE))" % (4 * c["SQ_ACTIVE_INST_VALU"] / (1024 * c["GRBM_GUI_ACTIVE"])))
    if "SQ_WAVE_CYCLES" in c and c["SQ_WAVE_CYCLES"] > 0:
        w = c["SQ_WAVE_CYCLES"]
        print("    of all wave-cycles: issuing %.3f, parked (s_waitcnt / barrier) %.3f, issue-stalled %.3f" % (
            c.get("SQ_ACTIVE_INST_ANY", 0) / w, c.get("SQ_WAIT_ANY", 0) / w, c.get("SQ_WAIT_INST_ANY", 0) / w))
    if "SQ_INSTS_VALU" in c and "GRBM_GUI_ACTIVE" in c:
        print("    VALU instructions / SIMD-cycle  %.4f" % (c["SQ_INSTS_VALU"] / (1024 * c["GRBM_GUI_ACTIVE"])))
