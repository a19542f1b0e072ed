#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
# every kernel alone on the GPU: bench.py --no-overlap (rvseg_schedule.overlap_build / overlap_layers = 0)
rm -rf $R/gpurun_out/pmc_fe; mkdir -p $R/gpurun_out/pmc_fe
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU" "TA_TA_BUSY_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout -k 10 150 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/pmc_fe/p$i -- python3 $R/bench.py --steps 1 --warmup 1 --no-overlap --no-cpu --no-latency --no-verify --extras none > $R/gpurun_out/pmc_fe/log$i.txt 2>&1
  echo "pass $i rc=$?"
done
python3 - <<'PY'
import csv,glob,collections,os,json
R=os.environ.get("GRAFT_REPO_ROOT","/root/repo")
agg=collections.defaultdict(lambda: collections.defaultdict(list))
dur=collections.defaultdict(list)
for f in sorted(glob.glob(R+"/gpurun_out/pmc_fe/p*/*/*counter_collection.csv")):
    per=collections.defaultdict(lambda: collections.defaultdict(float)); nm={}
    for r in csv.DictReader(open(f)):
        per[r["Dispatch_Id"]][r["Counter_Name"]]+=float(r["Counter_Value"]); nm[r["Dispatch_Id"]]=r["Kernel_Name"].split("(")[0].replace("void ","")
    for d,c in per.items():
        for k,v in c.items(): agg[nm[d]][k].append(v)
for f in sorted(glob.glob(R+"/gpurun_out/pmc_fe/p*/*/*kernel_trace.csv")):
    for r in csv.DictReader(open(f)):
        dur[r["Kernel_Name"].split("(")[0].replace("void ","")].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
out={}
for k,c in agg.items():
    if not k.startswith("rvseg::"): continue
    o={n:sum(v)/len(v) for n,v in c.items()}
    o["avg_us"]=sum(dur[k])/len(dur[k]) if dur.get(k) else 0
    out[k]=o
json.dump(out,open(R+"/gpurun_out/pmc_fe.json","w"),indent=1,sort_keys=True)
for k,o in sorted(out.items(), key=lambda kv:-kv[1]["avg_us"])[:14]:
    wc=o.get("SQ_WAVE_CYCLES",1)
    print("%-44s %7.0f us  wait %.2f  valu %.2f lds %.2f  instr/wavecyc valu %.3f  vmem %d lds %d  TAbusy/CU %.2f" % (k[7:51], o["avg_us"], o.get("SQ_WAIT_ANY",0)/wc, o.get("SQ_ACTIVE_INST_VALU",0)/max(1,o.get("SQ_BUSY_CYCLES",1)), o.get("SQ_ACTIVE_INST_LDS",0)/max(1,o.get("SQ_BUSY_CYCLES",1)), o.get("SQ_INSTS_VALU",0)/wc, o.get("SQ_INSTS_VMEM_RD",0), o.get("SQ_INSTS_LDS",0), o.get("TA_TA_BUSY_sum",0)/256/(o["avg_us"]*2400) if o["avg_us"] else 0))
PY
