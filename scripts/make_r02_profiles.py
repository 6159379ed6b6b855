#!/usr/bin/env python3
"""Copies the evidence of scripts/evidence_r02.sh (gpurun_out/ev2/) into profiles/ and renders the PMC counter log of the
conv shapes as profiles/r02_pmc_mfma.md.  Run in the build container after the gpurun call."""
import ast, json, os, re, shutil, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
E = os.path.join(R, "gpurun_out", "ev2")
P = os.path.join(R, "profiles")
subprocess.check_call([sys.executable, os.path.join(R, "scripts", "summarize_rocprof.py"), os.path.join(E, "r02_kernel_stats_graph.csv"), "12",
                       os.path.join(P, "r02_kernel_stats_graph.md")])
for f in ("r02_kernel_stats_graph.csv", "r02_bench_n1.json"):
    shutil.copy(os.path.join(E, f), os.path.join(P, f))
shutil.copy(os.path.join(E, "pmc_traffic.json"), os.path.join(P, "pmc_traffic.json"))
log = open(os.path.join(E, "pmc_conv.log")).read()
rows, raw = [], []
for b in log.split("== ")[1:]:
    lines = b.strip().split("\n")
    title = lines[0]
    cnt = {}
    for l in lines[1:]:
        m = re.match(r"(p[12]) (.*?) dispatches (\d+) (\{.*\})", l)
        if not m:
            continue
        name = m.group(2)
        short = ("conv_pws_kernel" if "conv_pws" in name else "conv_pw_kernel" if "conv_pw" in name else
                 "conv_wgrad_big_kernel" if "wgrad_big" in name else "conv_big_fixup_kernel" if "fixup" in name else
                 "conv_big_kernel" if "conv_big_kernel" in name else name)
        d = ast.literal_eval(m.group(4))
        cnt.setdefault(short, {}).update({k: float(v) for k, v in d.items()})
        raw.append(f"{title:36s} {short:24s} " + " ".join(f"{k} {v}" for k, v in d.items()))
    want = ("conv_wgrad_big_kernel" if title.endswith("wgrad") else "conv_pws_kernel" if "conv_pws_kernel" in cnt else
            "conv_pw_kernel" if "conv_pw_kernel" in cnt else "conv_big_kernel")
    c = cnt[want]
    cyc = c["SQ_BUSY_CYCLES"] / 32
    rows.append((title, want, c["SQ_BUSY_CYCLES"], cyc, c["SQ_INSTS_MFMA"], c["SQ_VALU_MFMA_BUSY_CYCLES"],
                 c["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * cyc), c["SQ_LDS_IDX_ACTIVE"] / 256 / cyc, c["SQ_LDS_BANK_CONFLICT"],
                 c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"]))
util = {r[0]: 100 * r[6] for r in rows}
out = ["# Round 2 - MFMA / LDS utilisation of the conv kernels at the metric's shapes (B = 32, 65 x 65 maps), from PMC counters (MI355X)\n",
       "Command (GPU box, `scripts/evidence_r02.sh`): per shape `bash scripts/pmc_conv.sh <tag> H W Cin Cout k s p d B mode iters` = two\n"
       "`rocprofv3 --pmc ... --kernel-trace` passes over `scripts/conv_one.py` (counters in their own runs, no other trace domain).  Values are per\n"
       "dispatch, summed over the chip (32 shader engines for `SQ_BUSY_CYCLES`, 1024 SIMDs for `SQ_VALU_MFMA_BUSY_CYCLES`, 256 CUs for\n"
       "`SQ_LDS_IDX_ACTIVE`).  MFMA pipe utilisation = MFMA_BUSY / (1024 SIMDs x kernel cycles), kernel cycles = SQ_BUSY_CYCLES / 32;\n"
       "`SQ_VALU_MFMA_BUSY_CYCLES` = 32 x `SQ_INSTS_MFMA` for `v_mfma_f32_32x32x16_bf16` (the weight-gradient kernel), 16 x for\n"
       "`v_mfma_f32_16x16x32_bf16` (`conv_big_kernel` since the 16x16 port: twice the instructions, the same busy cycles).\n",
       "| layer (H W Cin Cout k s p d B pass) | kernel | SQ_BUSY_CYCLES | kernel cycles | SQ_INSTS_MFMA | MFMA_BUSY | **MFMA pipe utilisation** | LDS array busy | LDS bank conflicts | waves parked (WAIT_ANY / WAVE_CYCLES) |\n|---|---|---:|---:|---:|---:|---:|---:|---:|---:|"]
for r in rows:
    out.append(f"| {r[0]} | `{r[1]}` | {r[2]:.3g} | {r[3]:.3g} | {r[4]:.3g} | {r[5]:.3g} | **{100*r[6]:.1f} %** | {100*r[7]:.1f} % | {r[8]:.2g} | {100*r[9]:.0f} % |")
u = [round(r[6] * 100, 1) for r in rows]
out.append(f"""
Reading.  The 3x3 stack the north-star target names holds the matrix pipes {u[2]:.0f}-{u[0]:.0f} % of the cycles (atrous 2048->256 {u[0]} %, 512->512 {u[1]} %,
256->256 {u[2]} %; weight gradient of the atrous layer {u[5]} %).  The chip lowers its clock under these bf16 MFMA loops (kernel cycles / wall
time; nominal 2.4 GHz: MI355X_MICROARCH.md 'DVFS give-back'), so `roofline.frac` and `roofline.stack_3x3.frac` in the bench line are priced
against the nominal peak, this table against the cycles the chip actually ran.

**The three forms of `conv_big_kernel`'s K loop this round, same three forward layers, same counters** (atrous 2048->256 / 512->512 / 256->256):
32x32x16 MFMAs, two wave groups half a phase apart (round 1): kernel cycles 1.62e6 / 9.38e5 / 2.88e5, pipe utilisation 68.5 / 64.8 / 52.9 %, waves
parked 47 %, wall 985 / 541 / 167.8 us;  16x16x32 (`EESEG_OPT_CONV_MFMA16`): 1.77e6 / 1.01e6 / 3.05e5 cycles, 62.8 / 60.4 / 49.9 %, parked 46 %, wall
924 / 512 / 161.8 us;  16x16x32 software-pipelined, eight waves in lockstep (`EESEG_OPT_CONV_SWP`, this table): {rows[0][3]:.3g} / {rows[1][3]:.3g} / {rows[2][3]:.3g} cycles,
{u[0]} / {u[1]} / {u[2]} %, parked {100*rows[0][9]:.0f} %, wall ~930 / 517 / 155.5 us (`scripts/m16_bench.py`).
The 16x16 shape needs ~9 % MORE cycles than 32x32 (twice the MFMA instructions: `SQ_ACTIVE_INST_ANY` 1.64e8 -> 1.95e8) and still finishes 3-7 %
EARLIER: the clock the chip holds under it is that much higher (1.62e6 cycles / 985 us = 1.65 GHz vs 1.77e6 / 924 us = 1.92 GHz on the atrous layer) - the
guide's 'DVFS give-back' item 7, and the reason cycle counts alone would have rejected the change.  The pipelined loop parks far less (`SQ_WAIT_ANY`
3.8e8 -> 2.4e8) and stalls more at MFMA issue (`SQ_WAIT_INST_ANY` 2.5e8 -> 3.4e8): two very different schedules, the same ~64 % of the pipe cycles - what
bounds the loop is the power-held clock and the issue port two MFMA-dense waves share, not where the waits sit.
256->256 loses its 15 points to tile quantisation (32 x 65 x 65 pixels = 528.1 tiles of 256 on 256 CUs: two rounds + a K-split tail + the
fix-up) and to the 7.5 us per-round fixed cost on a 36-K-tile loop.  The 1x1 layers are HBM-bound (SURVEY 8d): their figure of merit is
bytes/s, not MFMA cycles - `roofline.by_shape` in the bench line prices every layer against its own max(flops/peak, bytes/BW).  The
weight-stationary kernel (256->1024) shows 4 % of its LDS cycles as bank conflicts: its staging writes (8 bytes per lane into 512-byte rows,
16-byte chunks XOR-ed with the pixel) pair two lanes per bank group; its K loop reads are conflict-free.

Raw counter lines (per dispatch):
```""")
out += raw
out.append("```")
open(os.path.join(P, "r02_pmc_mfma.md"), "w").write("\n".join(out) + "\n")
d = json.load(open(os.path.join(P, "r02_bench_n1.json")))
r = d["roofline"]
print(d["value"], d["ms_per_step"], r["kernel"], r["achieved"], r["frac"], r["avg_launch_us"], r["traffic"], r["stack_3x3"])
print(d["secondary"]); print(d["cpu_baseline"])
print("\n".join(out[6:14]))
