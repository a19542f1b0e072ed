"""Cost model of a band-interleaved, block-resident ordered splat: every block of a frame owns
a fixed set of 7-vertex groups and walks them band by band, so the 7 readers of a pixel's row pass within a few
bands of each other.  Input: /tmp/off0.npy (N x 7 vertex ids of one bench frame, written by lattice_stats.py)."""
import sys
import numpy as np
off = np.load('/tmp/off0.npy'); N = off.shape[0]; M = int(off.max()) + 1
CYC_ADD, CYC_TILE, CYC_PROD = 8.5, 80.0, 400.0
GHZ = 2.4e9

def profiles(Bp):
    nb = (N + Bp - 1) // Bp
    c = np.zeros((M, nb), np.int64)
    band = np.repeat(np.arange(N) // Bp, 7)
    np.add.at(c, (off.ravel(), band), 1)
    return c

def piece_cost(n):          # n: array of max entries per (group, band)
    tiles = np.ceil(n / 64.0)
    full = np.floor(n / 64.0)
    rem = n - 64 * full
    cost = full * np.maximum(CYC_ADD * 64 + CYC_TILE, CYC_PROD)
    cost += (rem > 0) * np.maximum(CYC_ADD * np.ceil(rem / 4) * 4 + CYC_TILE, CYC_PROD)
    return cost, tiles

def groups_sorted(c):
    tot = c.sum(1); order = np.argsort(-tot); order = order[tot[order] > 0]
    return [order[i:i + 7] for i in range(0, len(order), 7)]

def groups_cooc(c):
    tot = c.sum(1); left = [v for v in np.argsort(-tot) if tot[v] > 0]
    out = []
    while left:
        seed = left.pop(0); g = [seed]; env = c[seed].copy()
        while len(g) < 7 and left:
            # the vertex whose profile adds least to the group's envelope, relative to its own mass
            best, bi = None, -1
            for i, v in enumerate(left[:60]):
                inc = np.maximum(env, c[v]).sum() - env.sum()
                score = inc / max(1, tot[v])
                if best is None or score < best: best, bi = score, i
            v = left.pop(bi); g.append(v); env = np.maximum(env, c[v])
        out.append(np.array(g))
    return out

for Bp in (1024, 2048, 4096, 8192):
    c = profiles(Bp)
    for name, gf in (("len-sorted", groups_sorted), ("co-occurrence", groups_cooc)):
        G = gf(c)
        env = np.stack([c[g].max(0) for g in G])           # groups x bands
        cost, tiles = piece_cost(env)
        gcost = cost.sum(1)
        for B in (8, 12, 16):
            # LPT packing of groups onto B blocks
            load = np.zeros(B); per_band = np.zeros((B, env.shape[1]))
            for gi in np.argsort(-gcost):
                b = int(np.argmin(load)); load[b] += gcost[gi]; per_band[b] += cost[gi]
            strict = per_band.max(0).sum()
            print("band %5d px %-13s B=%2d: groups %d tiles %d (adds %.2f N) | max block %.0f kcyc = %.3f ms, strict band sync %.3f ms, heavy group alone %.3f ms"
                  % (Bp, name, B, len(G), tiles.sum(), env.sum() / N, load.max() / 1e3, load.max() / GHZ * 1e3, strict / GHZ * 1e3, gcost.max() / GHZ * 1e3))
