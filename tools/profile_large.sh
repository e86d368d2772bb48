#!/bin/bash
# usage (on the GPU box, from the repo root): tools/profile_large.sh <tag>
# one large bundle-adjustment component (the wide point-major group): rocprofv3 kernel trace + stats, separate PMC passes
# (FETCH_SIZE / WRITE_SIZE: traffic.json; SQ counters), and the grid solver it replaces for comparison
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
D=gpurun_out/prof_$1
mkdir -p $D
B="python bench.py --workload large-component --steps 3 --warmup 1 --no-cpu-baseline"
for shape in 64x2000000x4 120x300000x4 356x226730x6; do
  T=large_$shape
  rocprofv3 --kernel-trace --stats --output-format csv -d $D -o $T -- $B --large-shape $shape > $D/bench_$T.json 2>> $D/err.txt
  mkdir -p $D/$T
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $D/$T -o fetch -- $B --large-shape $shape > /dev/null 2>> $D/err.txt
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $D/$T -o write -- $B --large-shape $shape > /dev/null 2>> $D/err.txt
  python tools/collect_traffic.py $D/$T large-component-$shape cgd_ptmg_kernel
done
T=large_64x2000000x4
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $D/$T -o valu -- $B > /dev/null 2>> $D/err.txt
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $D/$T -o wait -- $B > /dev/null 2>> $D/err.txt
python tools/pmc_summary.py $D/$T cgd_ptmg > $D/${T}_sq_summary.txt 2>> $D/err.txt
# the grid solver on the same component (plan option ptm_stream = 0)
rocprofv3 --kernel-trace --stats --output-format csv -d $D -o large_grid -- $B --opt ptm_stream=0 > $D/bench_large_grid.json 2>> $D/err.txt
cp profiles/traffic.json $D/
for f in $D/large_*_kernel_stats.csv; do echo $f; head -3 $f | cut -c1-200; done
cat $D/${T}_sq_summary.txt
tail -3 $D/err.txt
