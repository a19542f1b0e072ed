#!/bin/bash
# usage (on the GPU box): profiles/scripts/prof.sh <tag> [extra bench args]
# kernel trace + per-kernel statistics of 2 batch-64 steps of bench.py -> gpurun_out/prof_<tag>/
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=$1; shift
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$tag -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu --no-latency --no-verify --extras none "$@" > $R/gpurun_out/prof_$tag.log 2>&1
echo "prof rc=$?"
tail -1 $R/gpurun_out/prof_$tag.log | cut -c1-300
