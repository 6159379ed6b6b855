#!/usr/bin/env python3
"""Condense a rocprofv3 --kernel-trace --stats kernel_stats.csv into a small table.
usage: summarize_rocprof.py <kernel_stats.csv> <steps profiled> [out.md]"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2])
out = open(sys.argv[3], "w") if len(sys.argv) > 3 else sys.stdout
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    m = re.match(r"_ZN12_GLOBAL__N_1\d+([a-z_0-9]+?)I(DF16b|f)(?:Li(\d+))?", n)
    if m:
        return f"{m.group(1)}<{'bf16' if m.group(2) == 'DF16b' else 'f32'}{',' + m.group(3) if m.group(3) else ''}>"
    return re.sub(r"\(.*", "", n)[:70]
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"GPU busy per step: {tot / steps / 1e6:.2f} ms over {steps:g} steps, {sum(int(r['Calls']) for r in rows) / steps:.0f} launches/step\n", file=out)
print("| kernel | calls/step | ms/step | % | avg us |\n|---|---:|---:|---:|---:|", file=out)
# one 256x256 conv call = ONE conv_big_kernel<MODE> launch (<0> whole tiles only, <2> K-split tail blocks + whole tiles,
# <1> K ranges only when there is no full round) [+ conv_big_fixup_kernel]: report the per-call aggregate as well (what
# bench.py's HIP events bracket); calls = launches of conv_big_kernel<*>
big = [r for r in rows if "conv_big" in r["Name"]]
if big:
    calls = sum(int(r["Calls"]) for r in big if "conv_big_kernel" in r["Name"])
    tns = sum(float(r["TotalDurationNs"]) for r in big)
    if calls:
        print(f"| conv_big_kernel<bf16,256x256> per CALL (sum of the {len(big)} kernels below / launches of conv_big_kernel<*>) | {calls / steps:.1f} | "
              f"{tns / steps / 1e6:.3f} | {100 * tns / tot:.1f} | {tns / calls / 1e3:.1f} |", file=out)
for r in rows[:32]:
    print(f"| {short(r['Name'])} | {int(r['Calls']) / steps:.1f} | {float(r['TotalDurationNs']) / steps / 1e6:.3f} | "
          f"{float(r['Percentage']):.1f} | {float(r['AverageNs']) / 1e3:.1f} |", file=out)
