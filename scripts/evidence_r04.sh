#!/bin/bash
# Round-4 evidence on ONE box (run on the GPU box from the repo root): bench (default flags = the metric's workload, incl.
# secondary + CPU baseline), rocprofv3 kernel traces of the graph-replayed step at the metric's batch and at the per-GPU
# shards of the 8- and 4-GPU runs (4 and 8 images), PMC traffic passes (FETCH_SIZE / WRITE_SIZE, separate), MFMA utilisation
# of the six conv shapes of the round-2/3 tables + the small-M kernel.  Everything lands under gpurun_out/ev4/;
# scripts/make_r04_profiles.py renders profiles/r04_*.
R=$GRAFT_REPO_ROOT
E=$R/gpurun_out/ev4
mkdir -p $E
cd $R && python3 bench.py > $E/r04_bench_n1.json 2> $E/bench.err
echo "bench done: $(python3 -c "import json;d=json.load(open('$E/r04_bench_n1.json'));print(d['value'],d['ms_per_step'])")"
for B in 4 8; do
  python3 bench.py --batch-per-gpu $B --steps 30 --warmup 5 --no-cpu-baseline --no-secondary > $E/r04_bench_b$B.json 2> $E/bench_b$B.err
  echo "B=$B: $(python3 -c "import json;d=json.load(open('$E/r04_bench_b$B.json'));print(d['value'],d['ms_per_step'])")"
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $E/prof -o t -- python3 $R/bench.py --steps 8 --warmup 4 --no-cpu-baseline --no-secondary --no-kernel-events > $E/prof_bench.json 2> $E/prof.err
for B in 4 8; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $E/prof_b$B -o t -- python3 $R/bench.py --batch-per-gpu $B --steps 8 --warmup 4 --no-cpu-baseline --no-secondary --no-kernel-events > $E/prof_bench_b$B.json 2> $E/prof_b$B.err
done
cd $R
# each profiled process ran 2 eager warm-up steps, the capture pass (not executed) and the capture call's replay + 1 + 8 replays = 12 executed steps
cp $(find $E/prof -name "*kernel_stats.csv" | head -1) $E/r04_kernel_stats_graph.csv
python3 scripts/summarize_rocprof.py $E/r04_kernel_stats_graph.csv 12 $E/r04_kernel_stats_graph.md
for B in 4 8; do
  cp $(find $E/prof_b$B -name "*kernel_stats.csv" | head -1) $E/r04_kernel_stats_b$B.csv
  python3 scripts/summarize_rocprof.py $E/r04_kernel_stats_b$B.csv 12 $E/r04_kernel_stats_b$B.md
done
rm -rf $E/prof $E/prof_b4 $E/prof_b8
head -n 12 $E/r04_kernel_stats_graph.md
bash $R/scripts/pmc_traffic.sh > $E/pmc.log 2>&1 && cp $R/gpurun_out/pmc_traffic/pmc_traffic.json $E/pmc_traffic.json
tail -n 4 $E/pmc.log
i=0
: > $E/pmc_conv.log
for args in "65 65 2048 256 3 1 12 12 32 fwd" "65 65 512 512 3 1 4 4 32 fwd" "65 65 256 256 3 1 2 2 32 fwd" "65 65 256 1024 1 1 0 1 32 fwd" "65 65 1024 256 1 1 0 1 32 fwd" "65 65 2048 256 3 1 12 12 32 wgrad" "65 65 256 256 3 1 2 2 32 wgrad" "65 65 256 256 3 1 2 2 4 fwd" "65 65 1024 256 1 1 0 1 4 fwd"; do
  i=$((i+1))
  echo "== $args" >> $E/pmc_conv.log
  bash $R/scripts/pmc_conv.sh ev4_$i $args 3 >> $E/pmc_conv.log 2>&1
done
tail -n 20 $E/pmc_conv.log
