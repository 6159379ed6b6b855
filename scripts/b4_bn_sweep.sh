#!/bin/bash
# BatchNorm launch shapes at the 4-image shard: rows in flight per thread (EESEG_OPT_BN_ROWS 10, EESEG_OPT_BN_BWD_ROWS 18),
# grid cap of the column-fixed passes, blocks of the column reductions (EESEG_OPT_COLREDUCE_BLOCKS 11).
R=$GRAFT_REPO_ROOT
B=${1:-4}
mkdir -p $R/gpurun_out/r3
cd $R
run() { name=$1; shift; python3 bench.py --batch-per-gpu $B --no-cpu-baseline --no-secondary --no-kernel-events --steps 30 "$@" > gpurun_out/r3/bn_${B}_$name.json 2>/dev/null || { echo "$name failed"; return; }
  python3 -c "
import json
d=json.load(open('gpurun_out/r3/bn_${B}_$name.json')); print('$name', round(d['value'],1), 'img/s', round(d['ms_per_step'],2), 'ms')"
}
run base
run bwdrows2 --opt 18=2
run bwdrows4 --opt 18=4
run rows4 --opt 10=4
run rows1 --opt 10=1
run cap1024 --ew-grid-cap 1024
run cap2048 --ew-grid-cap 2048
run cap256 --ew-grid-cap 256
run colred1024 --opt 11=1024
run colred256 --opt 11=256
