#!/bin/bash
# A/B of the conv dispatch at the per-GPU shard of the 8-GPU run (4 images): 256-tile kernel + K-split tail + fix-up
# (default) vs the 128-tile kernel everywhere (EESEG_OPT_CONV_PIPE=0), with / without the 128x64 auto-narrow rule.
# usage (GPU box): bash scripts/b4_dispatch_ab.sh [batch per gpu]
R=$GRAFT_REPO_ROOT
B=${1:-4}
mkdir -p $R/gpurun_out/r3
cd $R
run() { name=$1; shift; python3 bench.py --batch-per-gpu $B --no-cpu-baseline --no-secondary --steps 20 "$@" > gpurun_out/r3/ab_${B}_$name.json 2> gpurun_out/r3/ab_${B}_$name.err || { tail -5 gpurun_out/r3/ab_${B}_$name.err; return 1; }
  python3 - <<PY
import json
d=json.load(open("gpurun_out/r3/ab_${B}_$name.json"))
print("$name", round(d["value"],1), "img/s", round(d["ms_per_step"],2), "ms", {k: round(v["ms_per_step"],2) for k,v in d["roofline"]["families"].items() if v["ms_per_step"]>0.3})
PY
}
run base
run pipe0 --opt 1=0
run pipe0_narrow --opt 1=0 --opt 4=1
run pw_all --opt 15=1
run nopws --opt 14=0
