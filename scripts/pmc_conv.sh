#!/bin/bash
# usage: pmc_conv.sh <tag> <conv_one args...>   (run on the GPU box, from the repo root)
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $out/p1 -- python3 $GRAFT_REPO_ROOT/scripts/conv_one.py "$@" > $out/p1.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM --kernel-trace --output-format csv -d $out/p2 -- python3 $GRAFT_REPO_ROOT/scripts/conv_one.py "$@" > $out/p2.log 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<PY
import csv, glob, collections
for p in ("p1", "p2"):
    f = glob.glob("$out/%s/**/*counter_collection.csv" % p, recursive=True)
    if not f:
        print(p, "no counter file", open("$out/%s.log" % p).read()[-600:]); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"][:60]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] in ("SQ_WAVE_CYCLES", "SQ_INSTS_LDS"): cnt[k] += 1
    for k, v in agg.items():
        if "conv" in k:
            print(p, k, "dispatches", cnt[k], {a: f"{b / max(cnt[k],1):.3g}" for a, b in v.items()})
PY
