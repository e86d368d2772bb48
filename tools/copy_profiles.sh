#!/bin/bash
# usage (in the builder's container, after gpurun merged the files back): tools/copy_profiles.sh <tag> <prefix>
# copies the summaries of tools/profile_round.sh <tag> / tools/pmc_round.sh <tag> from gpurun_out/ into profiles/<prefix>_*
set -u
cd "$(dirname "$0")/.."
S=gpurun_out/prof_$1; Q=gpurun_out/pmc_$1; P=profiles/$2
cp $S/ladybug_kernel_stats.csv ${P}_pipe_ladybug_kernel_stats.csv
tail -1 $S/bench_ladybug.json > ${P}_pipe_ladybug_bench.json
tail -1 $S/bench_ladybug_full.json > ${P}_pipe_ladybug_bench_default_line.json
cp $S/synth_kernel_stats.csv ${P}_batch_synthS_kernel_stats.csv; tail -1 $S/bench_synth.json > ${P}_batch_synthS_bench.json
cp $S/synthL_kernel_stats.csv ${P}_batch_synthL_kernel_stats.csv; tail -1 $S/bench_synthL.json > ${P}_batch_synthL_bench.json
cp $S/strong_kernel_stats.csv ${P}_strong_kernel_stats.csv; tail -1 $S/bench_strong.json > ${P}_strong_bench.json
cp $S/kernels_kernel_stats.csv ${P}_other_kernels_kernel_stats.csv; cp $S/kernels.txt ${P}_other_kernels_hip_events.txt
for w in ladybug-full synthetic-S synthetic-L synthetic-L-1000; do
  for c in fetch write; do
    (head -1 $S/$w/${c}_counter_collection.csv; grep 'cgd_' $S/$w/${c}_counter_collection.csv) > ${P}_${w}_pmc_${c}.csv
  done
done
for c in fetch write; do
  (head -1 $S/eval-grad/${c}_counter_collection.csv; grep 'grad_fused_kernel' $S/eval-grad/${c}_counter_collection.csv) > ${P}_eval-grad_pmc_${c}.csv
done
cp $S/grad_kernel_stats.csv ${P}_eval_grad_kernel_stats.csv; grep -v rocprofv3 $S/grad_probe.txt | grep -v "HSA version\|output_stream" > ${P}_eval_grad_probe.txt
cp $S/traffic.json profiles/traffic.json
for t in synthL synthS strong8 strong1; do
  (tail -1 $Q/${t}_target.txt; cat $Q/${t}_summary.txt) > ${P}_pmc_sq_${t}.txt
done
ls -la ${P}_* | wc -l
