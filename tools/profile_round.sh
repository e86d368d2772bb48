#!/bin/bash
# usage (on the GPU box, from the repo root): tools/profile_round.sh <tag>
# rocprofv3 kernel-trace + stats for both bench workloads, separate PMC passes, traffic.json
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
D=gpurun_out/prof_$1
mkdir -p $D
python bench.py --steps 20 --warmup 3 > $D/bench_ladybug_full.json 2> $D/err.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $D -o ladybug -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-strong-scaling > $D/bench_ladybug.json 2>> $D/err.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $D -o synth -- python bench.py --workload synthetic-S --steps 10 --warmup 2 --no-cpu-baseline > $D/bench_synth.json 2>> $D/err.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $D -o synthL -- python bench.py --workload synthetic-L --steps 3 --warmup 1 --no-cpu-baseline > $D/bench_synthL.json 2>> $D/err.txt
for w in ladybug-full synthetic-S synthetic-L; do
  mkdir -p $D/$w
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $D/$w -o fetch -- python bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline --no-strong-scaling > /dev/null 2>> $D/err.txt
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $D/$w -o write -- python bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline --no-strong-scaling > /dev/null 2>> $D/err.txt
done
# the strong-scaling block's own launch (1000 components of ladybug's size): counter passes of its own
mkdir -p $D/synthetic-L-1000
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $D/synthetic-L-1000 -o fetch -- python bench.py --workload synthetic-L --components 1000 --steps 2 --warmup 1 --no-cpu-baseline --no-strong-scaling > /dev/null 2>> $D/err.txt
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $D/synthetic-L-1000 -o write -- python bench.py --workload synthetic-L --components 1000 --steps 2 --warmup 1 --no-cpu-baseline --no-strong-scaling > /dev/null 2>> $D/err.txt
# the public gradient entry point in its streaming regime (8.0e6 factors): kernel trace and counter passes
mkdir -p $D/eval-grad
rocprofv3 --kernel-trace --stats --output-format csv -d $D -o grad -- python tools/gpu_probe_grad.py > $D/grad_probe.txt 2>> $D/err.txt
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $D/eval-grad -o fetch -- python tools/gpu_probe_grad.py > /dev/null 2>> $D/err.txt
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $D/eval-grad -o write -- python tools/gpu_probe_grad.py > /dev/null 2>> $D/err.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $D -o strong -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-all-components > $D/bench_strong.json 2>> $D/err.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $D -o kernels -- python tools/profile_kernels.py > $D/kernels.txt 2>> $D/err.txt
python tools/collect_traffic.py $D/ladybug-full ladybug-full cgd_pipe_
python tools/collect_traffic.py $D/synthetic-S synthetic-S cgd_lds_kernel
python tools/collect_traffic.py $D/synthetic-L synthetic-L cgd_ptm
python tools/collect_traffic.py $D/synthetic-L-1000 synthetic-L-1000 cgd_ptm
python tools/collect_traffic.py $D/eval-grad eval-grad grad_fused_kernel
cp profiles/traffic.json $D/
tail -1 $D/bench_ladybug_full.json
head -3 $D/ladybug_kernel_stats.csv
head -2 $D/synth_kernel_stats.csv
head -2 $D/synthL_kernel_stats.csv
tail -1 $D/bench_synthL.json | cut -c1-400
