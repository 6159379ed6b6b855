#!/bin/bash
# K-split of the 256x256 conv kernel at the small per-GPU shards: K tiles per range at least (EESEG_OPT_CONV_SPLIT_MIN_K 12)
# and the CUs a launch counts on (EESEG_OPT_CONV_CUS 8; fewer CUs -> fewer, longer K ranges and less slab traffic).
R=$GRAFT_REPO_ROOT
B=${1:-4}
mkdir -p $R/gpurun_out/r3
cd $R
run() { name=$1; shift; python3 bench.py --batch-per-gpu $B --no-cpu-baseline --no-secondary --no-kernel-events --steps 30 "$@" > gpurun_out/r3/split_${B}_$name.json 2>/dev/null || { echo "$name failed"; return; }
  python3 -c "
import json
d=json.load(open('gpurun_out/r3/split_${B}_$name.json')); print('$name', round(d['value'],1), 'img/s', round(d['ms_per_step'],2), 'ms')"
}
run base
run mink2 --opt 12=2
run mink6 --opt 12=6
run mink9 --opt 12=9
run mink12 --opt 12=12
run mink18 --opt 12=18
run cus192 --opt 8=192
run cus128 --opt 8=128
run base2
