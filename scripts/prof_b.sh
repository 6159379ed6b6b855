#!/bin/bash
# rocprofv3 kernel trace of the graph-replayed step at a given per-GPU batch: scripts/prof_b.sh <tag> <batch> [bench flags...]
# -> gpurun_out/r4/<tag>.md (scripts/summarize_rocprof.py; 12 executed steps = 2 eager + capture replay + 1 + 8)
R=$GRAFT_REPO_ROOT
TAG=$1; B=$2; shift 2
E=$R/gpurun_out/r4
mkdir -p $E
cd /tmp && export TMPDIR=/tmp
rm -rf $E/prof_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d $E/prof_$TAG -o t -- python3 $R/bench.py --batch-per-gpu $B --steps 8 --warmup 4 --no-cpu-baseline --no-secondary --no-kernel-events "$@" > $E/prof_$TAG.json 2> $E/prof_$TAG.err
cd $R
cp $(find $E/prof_$TAG -name "*kernel_stats.csv" | head -1) $E/$TAG.csv
python3 scripts/summarize_rocprof.py $E/$TAG.csv 12 $E/$TAG.md
rm -rf $E/prof_$TAG
head -n 40 $E/$TAG.md
