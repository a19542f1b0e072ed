import ctypes as C, numpy as np, torch, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import rovinasemanticsegmentation_amd as rv
from rovinasemanticsegmentation_amd import synthetic
W,H,N,n=640,480,640*480,64
dev=torch.device("cuda",0)
blob = synthetic.make_forest_bytes(seed=7, n_trees=4, leaves_per_tree=1 << 14, max_depth=30, single_classes=9, layer_classes=(8, 9))
rgb, depth = synthetic.make_batch(n); calib = synthetic.make_calib()
d_rgb=torch.from_numpy(rgb).to(dev); d_depth=torch.from_numpy(depth.view(np.int16)).to(dev)
d_marg=torch.empty((n,9*N),dtype=torch.float32,device=dev); d_lab=torch.empty((n,N),dtype=torch.int8,device=dev)
ctx=rv.Context(multi_layer=0,use_dense_crf=1,dcrf_iterations=5,label_mode=1,unknown_label=[8],max_batch=n,lattice_capacity_log2=12)
ctx.forest_load(blob)
s=torch.cuda.current_stream(dev).cuda_stream
for _ in range(2):
    ctx.segment_frames_device(n,d_rgb.data_ptr(),d_depth.data_ptr(),calib,0,d_marg.data_ptr(),d_lab.data_ptr(),s)
torch.cuda.synchronize()
L=ctx.L
L.rvseg_debug_queue.argtypes=[C.c_void_p]*8
cap=4_000_000
items=np.zeros((cap,4),np.uint32); tr=np.zeros((cap//7+1,4),np.uint64)
ntot=C.c_uint(); qb=(C.c_uint*8)(); qt=(C.c_uint*8)(); meta=(C.c_int*4)()
st=L.rvseg_debug_queue(ctx.h, items.ctypes.data_as(C.c_void_p), items.nbytes, C.byref(ntot), qb, qt, tr.ctypes.data_as(C.c_void_p), meta)
print("status",st,"groups",ntot.value,"meta",list(meta),"q_total",list(qt))
G=ntot.value
items=items[:G*7].reshape(G,7,4); tr=tr[:G]
# last splat launch's trace: times in 10 ns ticks
t0=tr[:,0][tr[:,0]>0].min()
pick=(tr[:,0]-t0)/100.0; deps=(tr[:,1]-t0)/100.0; done=(tr[:,2]-t0)/100.0; steps=(tr[:,3]>>np.uint64(32)).astype(np.int64); gate=(tr[:,3]&np.uint64(0xffffffff)).astype(np.float64)/100.0
print("launch span us: %.1f"%(done.max()))
q0=slice(qb[0], qb[0]+qt[0])
print("queue0 groups",qt[0])
# per-group breakdown in queue 0
w=(deps-pick)[q0]; r=(done-deps)[q0]; st_=steps[q0]
print("wait(pick->deps) us: mean %.2f p50 %.2f p90 %.2f max %.2f"%(w.mean(),np.median(w),np.percentile(w,90),w.max()))
print("run(deps->done) us: mean %.2f; steps mean %.2f; us/step %.3f"%(r.mean(), st_.mean(), r.sum()/st_.sum()))
print("gate(deps->first barrier) us mean %.2f"%(gate[q0].mean()))
# heavy chain of the frame slot 0 in queue 0: vertex of item 0 in the first group
v0=items[qb[0],0,0]
rows=[(g,i) for g in range(qb[0],qb[0]+qt[0]) for i in range(7) if items[g,i,0]==v0 and items[g,i,3]&1]
print("pieces of vertex",v0,len(rows))
prev=None
out=[]
for g,i in rows[:12]+rows[-3:]:
    out.append("g%d pick %.1f deps %.1f done %.1f steps %d tiles %d"%(g-qb[0],pick[g],deps[g],done[g],steps[g],(items[g,i,2]-items[g,i,1]+63)//64))
print("\n".join(out))
# busy fraction: sum of run time / (blocks * span)
print("block-time used for running: %.1f%% of 1024 x span"%(100*(done-deps).sum()/(1024*done.max())))
print("block-time waiting: %.1f%%"%(100*(deps-pick).sum()/(1024*done.max())))
