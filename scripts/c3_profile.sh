R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/c3
cd $R && python3 bench.py --arch resnet101 --branches 2 --classes 19 --batch-per-gpu 4 --no-cpu-baseline > gpurun_out/c3/bench.json 2> gpurun_out/c3/bench.err || exit 1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/c3/prof -o t -- python3 $R/bench.py --arch resnet101 --branches 2 --classes 19 --batch-per-gpu 4 --steps 6 --warmup 2 --no-cpu-baseline > $R/gpurun_out/c3/prof.json 2> $R/gpurun_out/c3/prof.err || exit 1
cd $R
f=$(find gpurun_out/c3/prof -name "*kernel_stats.csv" | head -1)
python3 scripts/summarize_rocprof.py $f 12 gpurun_out/c3/table.md > /dev/null
grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' gpurun_out/c3/bench.json | head -2
head -n 45 gpurun_out/c3/table.md
