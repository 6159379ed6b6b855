#!/bin/bash
# Which launches surround the __amd_rocclr_copyBuffer dispatches of a graph-replayed step?  (kernel trace of a short B=4 bench)
R=$GRAFT_REPO_ROOT
E=$R/gpurun_out/r4
mkdir -p $E
cd /tmp && export TMPDIR=/tmp
rm -rf $E/prof_cb
rocprofv3 --kernel-trace --output-format csv -d $E/prof_cb -o t -- python3 $R/bench.py --batch-per-gpu 4 --steps 3 --warmup 3 --no-cpu-baseline --no-secondary --no-kernel-events > $E/prof_cb.json 2> $E/prof_cb.err
cd $R
python3 - <<PY
import csv, glob, collections
f = glob.glob("$E/prof_cb/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
print(len(rows), "dispatches; columns:", list(rows[0].keys()))
# last third = graph replays; find copyBuffer and print neighbours + grid sizes
idx = [i for i, n in enumerate(names) if "copyBuffer" in n]
print("copyBuffer dispatches:", len(idx))
ctx = collections.Counter()
for i in idx[-120:]:
    prev = names[i - 1][:60] if i else "-"
    nxt = names[i + 1][:60] if i + 1 < len(names) else "-"
    g = rows[i].get("Grid_Size", rows[i].get("Grid_Size_X", "?"))
    ctx[(prev, nxt, g)] += 1
sgd = [i for i, n in enumerate(names) if "sgd_kernel" in n]
print("sgd_kernel dispatches:", len(sgd), "-> steps executed:", len(sgd) // 3)
after_first = [i for i in idx if i > sgd[0]]
print("copyBuffer dispatches after the first optimizer step:", len(after_first), "of", len(idx))
fills = [i for i, n in enumerate(names) if "FillFunctor" in n or "fillBuffer" in n]
print("fill dispatches:", len(fills), "after the first optimizer step:", len([i for i in fills if i > sgd[0]]))
last = rows[sgd[-1] - 3 if len(sgd) > 3 else 0]
# launches of the last replayed step = dispatches between the last two groups of three sgd_kernel launches
if len(sgd) >= 6:
    print("dispatches in the last replayed step:", sgd[-1] - sgd[-4])
for (p, n, g), c in ctx.most_common(8):
    print(f"{c:4d} x grid {g:>8}  after [{p}]  before [{n}]")
PY
rm -rf $E/prof_cb
