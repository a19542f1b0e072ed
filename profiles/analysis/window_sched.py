import numpy as np, sys
off = np.load('/tmp/off0.npy'); N = off.shape[0]; M = off.max()+1
cnt = np.bincount(off.ravel(), minlength=M)
def groups_by_length():
    order = np.argsort(-cnt)
    return [order[i:i+7] for i in range(0, M, 7)]
def groups_by_cooc(Bw=256):
    # co-occurrence at window granularity: cosine similarity of per-window count vectors
    nb = N//Bw
    H = np.zeros((M, nb), np.float32)
    for b in range(nb):
        c = np.bincount(off[b*Bw:(b+1)*Bw].ravel(), minlength=M)
        H[:, b] = c
    left = set(range(M)); groups=[]
    Hn = H / (np.linalg.norm(H,axis=1,keepdims=True)+1e-9)
    order = list(np.argsort(-cnt))
    while left:
        seed = next(v for v in order if v in left)
        g=[seed]; left.discard(seed)
        prof = H[seed].copy()
        while len(g)<7 and left:
            cand = np.array(sorted(left))
            # similarity to the group's summed profile
            pn = prof/ (np.linalg.norm(prof)+1e-9)
            sim = Hn[cand] @ pn
            v = int(cand[np.argmax(sim)])
            g.append(v); left.discard(v); prof += H[v]
        groups.append(np.array(g))
    return groups
def evaluate(groups, Bw):
    nb = N//Bw
    gid = np.zeros(M, int)
    for i,g in enumerate(groups): gid[g]=i
    steps=0; qbytes=0; pairs=0
    for b in range(nb):
        blk = off[b*Bw:(b+1)*Bw]
        c = np.bincount(blk.ravel(), minlength=M)
        t = (c+63)//64
        for i,g in enumerate(groups):
            m = t[g].max()
            if m>0:
                steps += m; pairs+=1
                # pixels of the window touched by any vertex of g
                touched = np.isin(blk, g).any(axis=1).sum()
                qbytes += touched*36
    return steps, qbytes/(N*36), pairs
for name,G in (("length",groups_by_length()),("cooc",groups_by_cooc())):
    for Bw in (64,256,1024,4096):
        st,q,p = evaluate(G,Bw)
        print("%-7s Bw %5d: block-steps %6d (list-sync 5315)  Q traffic %.2fx (vs 7x)  (group,window) pairs %d" % (name,Bw,st,q,p))
