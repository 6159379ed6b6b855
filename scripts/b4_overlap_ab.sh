#!/bin/bash
# Small per-GPU batches underfill the chip with every launch: does running the weight gradients on a side stream
# (beside the data-gradient = 1, beside the BatchNorm backward of the layer below = 2) recover some of it?
# usage (GPU box): bash scripts/b4_overlap_ab.sh [batch per gpu]
R=$GRAFT_REPO_ROOT
B=${1:-4}
mkdir -p $R/gpurun_out/r3
cd $R
run() { name=$1; shift; python3 bench.py --batch-per-gpu $B --no-cpu-baseline --no-secondary --no-kernel-events --steps 30 "$@" > gpurun_out/r3/ov_${B}_$name.json 2> gpurun_out/r3/ov_${B}_$name.err || { tail -5 gpurun_out/r3/ov_${B}_$name.err; return 1; }
  python3 -c "
import json
d=json.load(open('gpurun_out/r3/ov_${B}_$name.json'))
print('$name', round(d['value'],1), 'img/s', round(d['ms_per_step'],2), 'ms')"
}
run base
run overlap1 --overlap-wgrad 1
run overlap2 --overlap-wgrad 2
