#!/bin/bash
# usage (on the GPU box, from the repo root): tools/pmc_detail.sh <tag> [target]
# latency and queueing counters of a batch solver's kernel, four passes of four counters each (counters only, no tracing):
# LDS / vector-memory instructions in flight (level / instructions = cycles an instruction is in flight), instruction fetch,
# the issue cycles by instruction kind, LDS stalls
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
t=${2:-synthL}
D=gpurun_out/pmc_detail_$1
mkdir -p $D
i=0
for c in "SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL" \
         "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_IFETCH SQ_IFETCH_LEVEL" \
         "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" \
         "SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT SQ_INSTS_SALU SQ_INSTS_SMEM" \
         "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"; do
  i=$((i+1))
  RDIS_SYNTHL_COMPONENTS=256 rocprofv3 --pmc $c --output-format csv -d $D/$t -o p$i -- python tools/pmc_target.py $t >> $D/${t}_target.txt 2>> $D/err.txt
done
python tools/pmc_summary.py $D/$t cgd_ | tee $D/${t}_summary.txt
tail -2 $D/err.txt
