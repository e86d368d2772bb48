cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
D=gpurun_out/prof_r06_b; mkdir -p $D/large_120
B="python bench.py --workload large-component --steps 3 --warmup 1 --no-cpu-baseline --large-shape 120x300000x4"
rocprofv3 --kernel-trace --stats --output-format csv -d $D -o large_120 -- $B > $D/bench_large_120.json 2>> $D/err.txt
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $D/large_120 -o fetch -- $B > /dev/null 2>> $D/err.txt
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $D/large_120 -o write -- $B > /dev/null 2>> $D/err.txt
python tools/collect_traffic.py $D/large_120 large-component-120x300000x4 cgd_ptmg_kernel
cp profiles/traffic.json $D/
head -3 $D/large_120_kernel_stats.csv | cut -c1-220
tools/microbench/bin/eval_floor
python bench.py --steps 20 --warmup 5 > gpurun_out/bench_r06_b.json 2> gpurun_out/bench_r06_b.err; tail -c 1500 gpurun_out/bench_r06_b.json; tail -3 gpurun_out/bench_r06_b.err
