#!/bin/bash
# The N > 1 step of bench.py on ONE GPU: a 1-rank RCCL group forced to issue every collective (SyncBN statistics, CE
# count, gradient buckets), at the per-GPU batch of an N-GPU run.  A/B of EESEG_DEFER_WGRAD (weight gradients issued
# while the next SyncBN all-reduce is in flight).  usage (GPU box): bash scripts/dp_rehearsal_bench.sh [batch per gpu]
R=$GRAFT_REPO_ROOT
B=${1:-4}
mkdir -p $R/gpurun_out/dpr
cd $R
python3 bench.py --batch-per-gpu $B --no-cpu-baseline --no-secondary --no-kernel-events --steps 20 > gpurun_out/dpr/local_$B.json 2> gpurun_out/dpr/local_$B.err || exit 1
export EESEG_FORCE_ALLREDUCE=1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533
for d in 0 1; do
  EESEG_DEFER_WGRAD=$d python3 bench.py --gpus 1 --batch-per-gpu $B --no-cpu-baseline --no-secondary --no-kernel-events --steps 20 > gpurun_out/dpr/rccl_${B}_defer$d.json 2> gpurun_out/dpr/rccl_${B}_defer$d.err || exit 1
done
for f in local_$B rccl_${B}_defer0 rccl_${B}_defer1; do
  echo "$f $(grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"hip_graph": [a-z]*\|"sync_bn": [a-z]*' gpurun_out/dpr/$f.json | tr '\n' ' ')"
done
