#!/usr/bin/env python3
"""Joins the two runs of profiles/scripts/calibrate.sh into profiles/rNN_fetch_calibration.json.

usage: profiles/scripts/fetch_calibration.py gpurun_out/cal_<tag> profiles/rNN_fetch_calibration.json

Per kernel: the byte counts the program knows (useful bytes; bytes of the 64-B granules / 128-B lines
touched per wave-instruction), its un-profiled time, and FETCH_SIZE (KiB) of the profiled run (second
repetition of each kernel).  `factors` = what FETCH_SIZE (in bytes) has to be multiplied with to give
the bytes at 128-B line granularity -- the unit the fabric moves -- for each access shape."""
import collections
import csv
import glob
import json
import sys


def main():
    src, dst = sys.argv[1], sys.argv[2]
    plain = collections.OrderedDict()
    for line in open(src + "/plain.jsonl"):
        if line.startswith("{"):
            r = json.loads(line)
            plain[r["kernel"]] = r          # later repetitions overwrite the first (page-touching) one
    fetch = collections.defaultdict(list)
    for f in sorted(glob.glob(src + "/pmc/*/*counter_collection.csv")):
        per = collections.defaultdict(float)
        names = {}
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != "FETCH_SIZE":
                continue
            per[r["Dispatch_Id"]] += float(r["Counter_Value"])
            names[r["Dispatch_Id"]] = r["Kernel_Name"]
        for d in sorted(per, key=int):
            fetch[names[d]].append(per[d])
    out = {"kernels": {}, "factors": {}}
    for k, r in plain.items():
        cand = [v for name, v in fetch.items() if k.split("<")[0] in name and (("<" not in k) or (k.split("<")[1].rstrip(">") + ">" in name or "<" + k.split("<")[1] in name))]
        vals = cand[0] if cand else []
        # several index patterns reuse cal_gather36: dispatch order = program order
        o = dict(r)
        o["FETCH_SIZE_KiB_all_dispatches"] = vals
        out["kernels"][k] = o
    # cal_gather36 is dispatched 6 times (random x2, runs16 x2, runs64 x2), in that order
    g = next((v for name, v in fetch.items() if "cal_gather36" in name), [])
    for i, k in enumerate(["cal_gather36_random", "cal_gather36_runs16", "cal_gather36_runs64"]):
        if len(g) >= 2 * i + 2 and k in out["kernels"]:
            out["kernels"][k]["FETCH_SIZE_KiB"] = g[2 * i + 1]
    for k, o in out["kernels"].items():
        if "FETCH_SIZE_KiB" not in o and o["FETCH_SIZE_KiB_all_dispatches"]:
            o["FETCH_SIZE_KiB"] = o["FETCH_SIZE_KiB_all_dispatches"][-1]
        if "FETCH_SIZE_KiB" in o:
            fb = o["FETCH_SIZE_KiB"] * 1024.0
            o["fetch_over_useful"] = fb / o["useful_bytes"]
            o["fetch_over_64B_granules"] = fb / o["bytes_64B_granules"]
            o["fetch_over_128B_lines"] = fb / o["bytes_128B_lines"]
    def factor(k):
        o = out["kernels"].get(k, {})
        return round(1.0 / o["fetch_over_128B_lines"], 3) if o.get("fetch_over_128B_lines") else None
    out["factors"] = {"stream16": factor("cal_stream16"), "stream8": factor("cal_stream8"), "gather36": factor("cal_gather36_runs16"),
                      "gather36_random": factor("cal_gather36_random"), "gather36_runs64": factor("cal_gather36_runs64")}
    json.dump(out, open(dst, "w"), indent=1)
    print(json.dumps(out["factors"]))


if __name__ == "__main__":
    main()
