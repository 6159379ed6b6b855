#!/bin/bash
# HBM traffic of every kernel of one eager bench step (run on the GPU box from the repo root).
# Separate --pmc passes for FETCH_SIZE and WRITE_SIZE (TCC slots), kernel-trace only.
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_traffic
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/$c -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-graph --no-cpu-baseline --no-secondary --no-kernel-events > $out/$c.log 2>&1
done
cd $GRAFT_REPO_ROOT
python3 - <<PY
import csv, glob, json, collections, re
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("$out/%s/**/*counter_collection.csv" % c, recursive=True)
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] == c:
            a = agg[r["Kernel_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
    res[c] = agg
def merge(agg):
    m = collections.defaultdict(lambda: [0.0, 0])
    for k, (v, n) in agg.items():
        m[short(k)][0] += v; m[short(k)][1] += n
    return m
def short(n):
    m = re.match(r"_ZN12_GLOBAL__N_1\d+([a-z_0-9]+?)I(DF16b|f)(?:Li(\d+))?", n)
    if m:
        return f"{m.group(1)}<{'bf16' if m.group(2) == 'DF16b' else 'f32'}{',' + m.group(3) if m.group(3) else ''}>"
    return re.sub(r"\(.*", "", n.replace("(anonymous namespace)::", "").replace("void ", ""))[:60]
out = {}
res = {c: merge(a) for c, a in res.items()}      # template variants of one kernel family are summed
for k in res["FETCH_SIZE"]:
    fs, n = res["FETCH_SIZE"][k]; ws, n2 = res["WRITE_SIZE"].get(k, [0.0, 1])
    # units: KiB.  gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE counts half of wide coalesced reads -> x2
    out[k] = {"launches": n, "fetch_bytes_per_launch": 2 * fs * 1024 / n, "write_bytes_per_launch": ws * 1024 / max(n2, 1),
                     "hbm_bytes_per_launch": (2 * fs + ws * n / max(n2, 1)) * 1024 / n}
# one 256x256 conv CALL = one conv_big_kernel<MODE> launch [+ fix-up]: per-call aggregate under the family name bench.py reports
bigs = [k for k in out if k.startswith("conv_big")]
calls = sum(out[k]["launches"] for k in bigs if k.startswith("conv_big_kernel"))
if calls:
    out["conv_big_kernel<bf16,256x256>"] = {
        "launches": calls,
        "fetch_bytes_per_launch": sum(out[k]["fetch_bytes_per_launch"] * out[k]["launches"] for k in bigs) / calls,
        "write_bytes_per_launch": sum(out[k]["write_bytes_per_launch"] * out[k]["launches"] for k in bigs) / calls,
        "hbm_bytes_per_launch": sum(out[k]["hbm_bytes_per_launch"] * out[k]["launches"] for k in bigs) / calls}
json.dump(out, open("$out/traffic_raw.json", "w"), indent=1)
json.dump({"workload_key": "resnet101-3-513-32-bf16",
           "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) over one eager bench step; "
                     "KiB units; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts half of wide coalesced reads); "
                     "counts traffic beyond L2 (Infinity-Cache hits included)",
           "hbm_bytes_per_launch": {k: v["hbm_bytes_per_launch"] for k, v in out.items()},
           "fetch_bytes_per_launch": {k: v["fetch_bytes_per_launch"] for k, v in out.items()},
           "write_bytes_per_launch": {k: v["write_bytes_per_launch"] for k, v in out.items()},
           "launches": {k: v["launches"] for k, v in out.items()}},
          open("$out/pmc_traffic.json", "w"), indent=1)
for k, v in sorted(out.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"])[:14]:
    print(f"{k:40s} n={v['launches']:4d} fetch/launch {v['fetch_bytes_per_launch']/1e6:9.2f} MB  write/launch {v['write_bytes_per_launch']/1e6:9.2f} MB")
PY
