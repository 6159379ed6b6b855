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
for r in rows[:28]:
    print(f"| {short(r['Name'])} | {int(r['Calls']) / steps:.1f} | {float(r['TotalDurationNs']) / steps / 1e6:.3f} | "
          f"{float(r['Percentage']):.1f} | {float(r['AverageNs']) / 1e3:.1f} |", file=out)
