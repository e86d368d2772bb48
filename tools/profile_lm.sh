#!/bin/bash
# usage (on the GPU box, from the repo root): tools/profile_lm.sh <tag>
# Levenberg-Marquardt on full ladybug (tools/gpu_lm_profile.py): rocprofv3 kernel trace + stats, and a separate counter pass for the
# matrix cores (SQ_INSTS_MFMA, SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CYCLES), summed per kernel
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
D=gpurun_out/prof_$1
mkdir -p $D/lm
rocprofv3 --kernel-trace --stats --output-format csv -d $D -o lm -- python tools/gpu_lm_profile.py > $D/lm_probe.txt 2>> $D/err.txt
rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $D/lm -o mfma -- python tools/gpu_lm_profile.py > /dev/null 2>> $D/err.txt
python - "$D/lm/mfma_counter_collection.csv" > $D/lm_pmc_mfma.txt <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for row in csv.DictReader(open(sys.argv[1])):
    k = row["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].split("::")[-1]
    acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
    if row["Counter_Name"] == "SQ_BUSY_CYCLES": n[k] += 1
for k in sorted(acc, key=lambda k: -acc[k].get("SQ_INSTS_MFMA", 0)):
    d = max(n[k], 1)
    print("%s: dispatches %d, SQ_INSTS_MFMA %.4g, SQ_VALU_MFMA_BUSY_CYCLES %.4g, SQ_BUSY_CYCLES %.4g per dispatch" % (
        k, n[k], acc[k].get("SQ_INSTS_MFMA", 0) / d, acc[k].get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / d, acc[k].get("SQ_BUSY_CYCLES", 0) / d))
PY
head -8 $D/lm_kernel_stats.csv | cut -c1-160; cat $D/lm_pmc_mfma.txt | head -8; tail -2 $D/lm_probe.txt
