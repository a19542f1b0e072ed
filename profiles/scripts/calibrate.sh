#!/bin/bash
# usage (on the GPU box): profiles/scripts/calibrate.sh <tag>
# runs profiles/scripts/fetch_calibration plain (times) and under rocprofv3 --pmc FETCH_SIZE (counter)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/cal_$1
mkdir -p $O
$R/profiles/scripts/fetch_calibration > $O/plain.jsonl 2> $O/plain.err
echo "plain rc=$?"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc -- $R/profiles/scripts/fetch_calibration > $O/pmc.jsonl 2> $O/pmc.err
echo "pmc rc=$?"
