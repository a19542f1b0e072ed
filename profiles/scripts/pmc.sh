#!/bin/bash
# usage (on the GPU box): profiles/scripts/pmc.sh <tag>
# HBM counters of bench.py, one rocprofv3 --pmc pass per counter set (FETCH_SIZE needs 3 of the 4 TCC
# slots, WRITE_SIZE 2: MI355X_MICROARCH.md "rocprofv3 PMC slots"), each with --kernel-trace only.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/pmc_$1
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/pmc_$1/p$i -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu --no-latency --no-verify --extras none > $R/gpurun_out/pmc_$1/log$i.txt 2>&1
  echo "pass $i rc=$?"
done
