#!/bin/bash
# Round evidence on ONE box: bench (default flags), rocprofv3 kernel trace of the same command, PMC traffic passes.
# usage (GPU box, repo root): bash scripts/evidence.sh <tag>
tag=${1:-r02}
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/ev
bash $R/scripts/pmc_traffic.sh > $R/gpurun_out/ev/pmc.log 2>&1 && cp $R/gpurun_out/pmc_traffic/pmc_traffic.json $R/profiles/pmc_traffic.json
cd $R && python3 bench.py > gpurun_out/ev/${tag}_bench_n1.json 2> gpurun_out/ev/bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ev/prof -o t -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-secondary > $R/gpurun_out/ev/prof_bench.json 2> $R/gpurun_out/ev/prof.err
cd $R
f=$(find gpurun_out/ev/prof -name "*kernel_stats.csv" | head -1)
cp $f gpurun_out/ev/${tag}_kernel_stats_graph.csv
python3 scripts/summarize_rocprof.py $f 12 gpurun_out/ev/${tag}_kernel_stats_graph_table.md
tail -n 3 gpurun_out/ev/pmc.log
python3 -c "import json; d=json.load(open('gpurun_out/ev/${tag}_bench_n1.json')); print(d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['achieved'], d['roofline']['avg_launch_us'], d['roofline']['traffic'], d['cpu_baseline'])"
head -n 12 gpurun_out/ev/${tag}_kernel_stats_graph_table.md
