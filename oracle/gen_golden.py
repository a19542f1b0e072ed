#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY -- regenerates tests/golden/forest_* with the REFERENCE's own code.

Runs only in the build container (it needs oracle/_ref/libforest_ref, i.e. the reference's
third-party/libforest/src/classifier.cpp compiled in place by `make -C oracle ref`).  The forests
are serialised by the reference's RandomForest::write and the expected outputs are produced by
the reference's classLogPosterior / multiClassLogPosterior (classifier.cpp:166-235).  The vectors
pin the model format, the strict '<' split rule and the tree-order fp32 accumulation bit-exactly.
"""
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "..", "tests", "golden")
REF = os.path.join(HERE, "_ref", "libforest_ref")


def points(seed, P, D=366):
    rng = np.random.default_rng(seed)
    X = np.empty((P, D), np.float32)
    X[:, :363] = rng.integers(0, 256, (P, 363)).astype(np.float32)  # Lab bytes stored as float
    X[:, 363] = rng.uniform(0.5, 15.0, P)                           # depth [m]
    X[:, 364] = rng.uniform(-1.0, 3.0, P)                           # height [m]
    X[:, 365] = np.where(rng.random(P) < 0.1, -2.0, rng.uniform(0, np.pi / 2, P))  # normal angle
    return X


def main():
    if not os.path.exists(REF):
        sys.exit("oracle/_ref/libforest_ref missing: run `make -C oracle ref` in the build container")
    os.makedirs(GOLD, exist_ok=True)
    tmp = "/tmp/rvseg_golden"
    os.makedirs(tmp, exist_ok=True)
    X = points(20240, 1024)
    X.tofile(os.path.join(tmp, "x.f32"))
    # single-label forest (C=9) and shared two-layer forest (8, 9) that also carries
    # single-label histograms, both T=4
    specs = {
        "forest_single": ["11", "4", "96", "14", "366", "9", "0"],
        "forest_multi": ["12", "4", "96", "14", "366", "9", "2", "8", "9"],
        "forest_tiny": ["13", "2", "3", "2", "4", "2", "2", "3", "2"],
    }
    out = {"points": X}
    for name, args in specs.items():
        dat = os.path.join(GOLD, name + ".dat")
        subprocess.check_call([REF, "gen", dat] + args)
        D = int(args[4])
        if D == 366:
            xs = os.path.join(tmp, "x.f32")
        else:
            Xt = np.random.default_rng(5).uniform(0, 255, (64, D)).astype(np.float32)
            Xt[::3] = np.floor(Xt[::3])
            xs = os.path.join(tmp, "xt.f32")
            Xt.tofile(xs)
            out["points_tiny"] = Xt
        for mode in ("single", "multi"):
            if mode == "multi" and args[6] == "0":
                continue
            o = os.path.join(tmp, name + "_" + mode + ".f32")
            subprocess.check_call([REF, "eval", dat, xs, str(D), mode, o])
            out[name + "_" + mode] = np.fromfile(o, np.float32)
    np.savez_compressed(os.path.join(GOLD, "forest_vectors.npz"), **out)
    print("wrote", sorted(out.keys()))


if __name__ == "__main__":
    main()
