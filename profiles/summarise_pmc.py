#!/usr/bin/env python3
"""Summarises the rocprofv3 --pmc passes written by scratch/pmc2.sh (one pass per counter set:
FETCH_SIZE / WRITE_SIZE / TCC_HIT_sum+TCC_MISS_sum, each with --kernel-trace) into the small JSON
that bench.py reads for `roofline.traffic`.

usage: profiles/summarise_pmc.py <gpurun_out/pmc_TAG> <profiles/rNN_x_pmc_batch64.json>

Per kernel (name cut at the first '('): counters averaged over all dispatches of the run
(FETCH_SIZE / WRITE_SIZE stay in KiB as rocprofv3 reports them), the mean duration from the kernel
trace of the same passes, and launches per bench step (the profiled command runs 2 steps)."""
import collections
import csv
import glob
import json
import sys


def main():
    src, dst = sys.argv[1], sys.argv[2]
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in sorted(glob.glob(src + "/p*/*/*counter_collection.csv")):
        per_dispatch = collections.defaultdict(lambda: collections.defaultdict(float))
        names = {}
        for r in csv.DictReader(open(f)):
            per_dispatch[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
            names[r["Dispatch_Id"]] = r["Kernel_Name"].split("(")[0].replace("void ", "")
        for d, c in per_dispatch.items():
            for k, v in c.items():
                agg[names[d]][k].append(v)
    dur = collections.defaultdict(list)
    for f in sorted(glob.glob(src + "/p*/*/*kernel_trace.csv")):
        for r in csv.DictReader(open(f)):
            dur[r["Kernel_Name"].split("(")[0].replace("void ", "")].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    n_passes = max(1, len(glob.glob(src + "/p*/*/*kernel_trace.csv")))
    out = {}
    for k, c in sorted(agg.items()):
        if not k.startswith("rvseg::"):
            continue
        o = {name: sum(v) / len(v) for name, v in sorted(c.items())}
        if dur.get(k):
            o["avg_us_profiled"] = sum(dur[k]) / len(dur[k])
            o["launches_per_step"] = round(len(dur[k]) / n_passes / steps)
        out[k] = o
    json.dump(out, open(dst, "w"), indent=1, sort_keys=True)
    print("wrote", dst, len(out), "kernels")


if __name__ == "__main__":
    main()
