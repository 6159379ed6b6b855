#!/bin/bash
# kernel-trace profile of the metric's network at one per-GPU batch (the shard a rank of an N-GPU run sees):
#   prof_shape.sh <batch per gpu> <tag> [extra bench flags]     -> gpurun_out/<tag>/{bench.json,table.md}
R=$GRAFT_REPO_ROOT
B=$1; T=$2; shift 2
O=$R/gpurun_out/$T
mkdir -p $O
cd $R && python3 bench.py --batch-per-gpu $B --no-cpu-baseline --no-secondary --no-kernel-events "$@" > $O/bench.json 2> $O/bench.err || exit 1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o t -- python3 $R/bench.py --batch-per-gpu $B --steps 8 --warmup 4 --no-cpu-baseline --no-secondary --no-kernel-events "$@" > $O/prof.json 2> $O/prof.err || exit 1
cd $R
f=$(find $O/prof -name "*kernel_stats.csv" | head -1)
python3 scripts/summarize_rocprof.py $f 12 $O/table.md > /dev/null
grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' $O/bench.json | head -2
head -n 40 $O/table.md
