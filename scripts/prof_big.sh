#!/bin/bash
# kernel-trace of single conv shapes under the 256x256 kernel (EESEG_CONV_PIPE=3); run on the GPU box
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export EESEG_CONV_PIPE=3
i=0
for args in "65 65 2048 256 3 1 12 12 16 fwd" "65 65 2048 256 3 1 12 12 16 dgrad" "65 65 256 256 3 1 2 2 16 fwd" "65 65 512 512 3 1 2 2 16 fwd"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pb$i -o t -- python3 $R/scripts/conv_one.py $args 5 > /dev/null 2>&1
  echo "== $args"
  f=$(find $R/gpurun_out/pb$i -name "*kernel_stats.csv" | head -1)
  grep "conv_big\|conv_igemm" $f | cut -d, -f1-6
done
