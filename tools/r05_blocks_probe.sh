#!/bin/bash
# probe: do the exiting workgroups of the padded planes set the floor of the reducing loss kernels? (MG_LOSS_BLOCKS=64: a quarter of the workgroups per plane)
out=gpurun_out/r05j; mkdir -p $out; root=$(pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pr_stats
MG_LOSS_BLOCKS=64 timeout 40 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pr_stats -- python $root/bench.py --steps 6 --warmup 2 --no-roofline --no-cpu-baseline > /dev/null 2> /tmp/pr_stats.err
f=$(find /tmp/pr_stats -name '*kernel_stats.csv' | head -1)
grep -i "pyr_lap_fwd\|point_fwd_kernel" $f | awk -F, '{print $1, $(NF-6), $(NF-4), $(NF-2), $(NF-1)}' | cut -c1-140 | tee $root/$out/blocks64_stats.txt
