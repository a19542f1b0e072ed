import numpy as np
off = np.load('/tmp/off0.npy'); N = off.shape[0]; M = off.max()+1
for Bp in (64, 256, 1024, 2048, 4096, 8192, 16384):
    nb = N//Bp
    tot_max = 0; tot_tiles = 0; tot_steps7 = 0; items=0
    for b in range(nb):
        c = np.bincount(off[b*Bp:(b+1)*Bp].ravel(), minlength=M)
        c = c[c>0]
        tot_max += c.max()
        t = np.ceil(c/64).astype(int)
        tot_tiles += t.sum(); items += len(c)
        s = np.sort(t)[::-1]
        tot_steps7 += sum(s[i] for i in range(0,len(s),7))
    print("Bp %5d: sum_b max_v n = %.3f N (%d tiles); tiles %d (vs %d); steps with 7-groups %d; items/band %.1f" % (Bp, tot_max/N, tot_max//64, tot_tiles, N*7//64, tot_steps7, items/nb))
