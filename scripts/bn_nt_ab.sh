#!/bin/bash
# A/B of EESEG_OPT_BN_NT (nontemporal loads of the dead-after-this-pass inputs of bn_apply / bn_bwd_apply) on the bench step.
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r3
cd $R
for v in 0 1 2 3; do
  python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-kernel-events --opt 20=$v "$@" > gpurun_out/r3/nt_$v.json 2> gpurun_out/r3/nt_$v.err || { tail -3 gpurun_out/r3/nt_$v.err; continue; }
  python3 -c "
import json
d=json.load(open('gpurun_out/r3/nt_$v.json')); print('BN_NT=$v', round(d['value'],1), 'img/s', round(d['ms_per_step'],2), 'ms')"
done
