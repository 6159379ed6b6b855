#!/bin/bash
# Round-2 evidence on ONE box (run on the GPU box from the repo root): bench (default flags = the metric's workload),
# rocprofv3 kernel trace of the same command, PMC traffic passes (FETCH_SIZE / WRITE_SIZE, separate), MFMA utilisation of
# the four conv shapes VERDICT r1 names + the pointwise kernel.  Everything lands under gpurun_out/ev2/.
R=$GRAFT_REPO_ROOT
E=$R/gpurun_out/ev2
mkdir -p $E
cd $R && python3 bench.py > $E/r02_bench_n1.json 2> $E/bench.err
echo "bench done: $(python3 -c "import json;d=json.load(open('$E/r02_bench_n1.json'));print(d['value'],d['ms_per_step'])")"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $E/prof -o t -- python3 $R/bench.py --steps 8 --warmup 4 --no-cpu-baseline --no-secondary --no-kernel-events > $E/prof_bench.json 2> $E/prof.err
cd $R
f=$(find $E/prof -name "*kernel_stats.csv" | head -1)
cp $f $E/r02_kernel_stats_graph.csv
# the profiled process ran 2 eager warm-up steps, the capture pass (not executed) and the capture call's replay + 1 + 8 replays = 12 executed steps
python3 scripts/summarize_rocprof.py $f 12 $E/r02_kernel_stats_graph.md
head -n 16 $E/r02_kernel_stats_graph.md
bash $R/scripts/pmc_traffic.sh > $E/pmc.log 2>&1 && cp $R/gpurun_out/pmc_traffic/pmc_traffic.json $E/pmc_traffic.json
tail -n 4 $E/pmc.log
i=0
for args in "65 65 2048 256 3 1 12 12 32 fwd" "65 65 512 512 3 1 4 4 32 fwd" "65 65 256 256 3 1 2 2 32 fwd" "65 65 256 1024 1 1 0 1 32 fwd" "65 65 1024 256 1 1 0 1 32 fwd" "65 65 2048 256 3 1 12 12 32 wgrad"; do
  i=$((i+1))
  echo "== $args" >> $E/pmc_conv.log
  bash $R/scripts/pmc_conv.sh ev2_$i $args 3 >> $E/pmc_conv.log 2>&1
done
tail -n 30 $E/pmc_conv.log
