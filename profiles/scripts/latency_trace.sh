#!/bin/bash
# usage (on the GPU box): profiles/scripts/latency_trace.sh {frame|cloud|deep|batch} <tag>
# kernel trace of one latency shape -> timeline of the last repetition in gpurun_out/lat_<tag>.txt
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
what=$1; tag=$2
rm -rf $R/gpurun_out/lat_$tag
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/lat_$tag -- python3 $R/profiles/scripts/latency_shapes.py $what 10 > $R/gpurun_out/lat_$tag.log 2>&1
echo "trace rc=$?"
tail -3 $R/gpurun_out/lat_$tag.log | cut -c1-1500
python3 - $R/gpurun_out/lat_$tag $what > $R/gpurun_out/lat_$tag.txt <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/*/*kernel_trace.csv"):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last repetition: walk back from the end to the last prep_kernel (frame) / fusion kernel (cloud)
first = "fusion" if sys.argv[2] == "cloud" else "prep_kernel"
idx = [i for i, r in enumerate(rows) if first in r["Kernel_Name"]]
# cloud: several fusion kernels per call; take the first of the last group
start = idx[-1]
while start - 1 in idx or (sys.argv[2] == "cloud" and any(start - k in idx for k in range(1, 6))):
    start = max(j for j in idx if j < start and start - j <= 6)
sel = rows[start:]
t0 = int(sel[0]["Start_Timestamp"])
end = max(int(r["End_Timestamp"]) for r in sel)
print("last repetition: %d kernels, %.3f ms from first start to last end, sum of durations %.3f ms" % (
    len(sel), (end - t0) / 1e6, sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in sel) / 1e6))
prev_end = {}
for r in sel:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    q = r.get("Queue_Id", "?")
    gap = (s - prev_end[q]) / 1e3 if q in prev_end else 0.0
    prev_end[q] = e
    print("%9.1f us  +%7.1f us  gap %6.1f  q%-3s grid %-8s %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap, q, r.get("Grid_Size", "?"),
          r["Kernel_Name"].split("(")[0].replace("void ", "").replace("rvseg::", "")[:90]))
PY
head -3 $R/gpurun_out/lat_$tag.txt
