#!/bin/bash
# What planning the 256-tile kernels for fewer CUs costs at the per-GPU shards of the N = 8 / 4 / 2 runs (no collectives here: the cost
# side only; the benefit - no second round while RCCL kernels hold CUs - needs the multi-GPU node).
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r3
cd $R
run() { B=$1; name=$2; shift; shift; python3 bench.py --batch-per-gpu $B --no-cpu-baseline --no-secondary --no-kernel-events --steps 30 "$@" > gpurun_out/r3/rsv_${B}_$name.json 2>/dev/null || { echo "$B $name failed"; return; }
  python3 -c "
import json
d=json.load(open('gpurun_out/r3/rsv_${B}_$name.json')); print('B=$B $name', round(d['value'],1), 'img/s', round(d['ms_per_step'],2), 'ms')"
}
for B in 4 8 16; do
run $B cus256
run $B cus248 --opt 8=248 --wgrad-big-grid 248,8
run $B cus240 --opt 8=240 --wgrad-big-grid 240,8
run $B cus224 --opt 8=224 --wgrad-big-grid 224,8
done
