#!/bin/bash
# Where does the data-parallel path spend its extra time?  Kernel traces of the bench with and without a 1-rank RCCL
# group that is forced to issue every collective (EESEG_FORCE_ALLREDUCE=1).  usage (GPU box): bash scripts/dp_overhead.sh
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/dp
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/dp/local -o t -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline > $R/gpurun_out/dp/local.json 2> $R/gpurun_out/dp/local.err || exit 1
export EESEG_FORCE_ALLREDUCE=1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29531
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/dp/rccl -o t -- python3 $R/bench.py --gpus 1 --steps 6 --warmup 2 --no-cpu-baseline > $R/gpurun_out/dp/rccl.json 2> $R/gpurun_out/dp/rccl.err || exit 1
cd $R
for k in local rccl; do
  f=$(find gpurun_out/dp/$k -name "*kernel_stats.csv" | head -1)
  python3 scripts/summarize_rocprof.py $f 12 gpurun_out/dp/${k}_table.md > /dev/null
  grep -o '"value": [0-9.]*' gpurun_out/dp/$k.json | head -1
  head -n 3 gpurun_out/dp/${k}_table.md
done
