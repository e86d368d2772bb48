#!/bin/bash
# usage (on the GPU box, from the repo root): tools/pmc_round.sh <tag>
# SQ counter passes (VALU issue, wave stalls, LDS bank conflicts) of the batch solvers' kernels -- counters only, no tracing in the same run
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
D=gpurun_out/pmc_$1
mkdir -p $D
for t in synthL synthS strong8 strong1; do
  RDIS_SYNTHL_COMPONENTS=256 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $D/$t -o valu -- python tools/pmc_target.py $t > $D/${t}_target.txt 2>> $D/err.txt
  RDIS_SYNTHL_COMPONENTS=256 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $D/$t -o wait -- python tools/pmc_target.py $t >> $D/${t}_target.txt 2>> $D/err.txt
  RDIS_SYNTHL_COMPONENTS=256 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS --output-format csv -d $D/$t -o lds -- python tools/pmc_target.py $t >> $D/${t}_target.txt 2>> $D/err.txt
  python tools/pmc_summary.py $D/$t cgd_ > $D/${t}_summary.txt 2>> $D/err.txt
  cat $D/${t}_target.txt | tail -1; cat $D/${t}_summary.txt
done
tail -5 $D/err.txt
