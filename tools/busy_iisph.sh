#!/bin/bash
# VALU / SALU / TA busy and instructions per wave of the IISPH chain's kernels (bench.py --solver iisph --config C2, one PMC pass)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_iisph; mkdir -p $OUT
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INST_CYCLES_SALU SQ_INSTS_SALU SQ_WAVES TA_BUSY_avr SQ_WAVE_CYCLES -d $OUT/p -o p --output-format csv -- python3 $R/bench.py --solver iisph --config C2 --steps 6 --warmup 3 --no-cpu-baseline > $OUT/run.log 2>&1 || { tail -5 $OUT/run.log; exit 1; }
python3 - "$OUT" <<'PY'
import csv,sys,glob,collections,re
f=glob.glob("%s/p/**/*counter_collection.csv"%sys.argv[1],recursive=True)[0]
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    m=re.search(r"nrs::(k_\w+)",r["Kernel_Name"])
    if m: acc[m.group(1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
rows=[]
for nm,c in acc.items():
    m={k:sum(x)/len(x) for k,x in c.items()}
    gui=m["GRBM_GUI_ACTIVE"]/8.0
    rows.append((gui*len(c["GRBM_GUI_ACTIVE"]),nm,gui,100*m["SQ_ACTIVE_INST_VALU"]/256/gui,100*m["SQ_INST_CYCLES_SALU"]/256/gui,100*m["TA_BUSY_avr"]/gui,m["SQ_INSTS_VALU"]/max(1,m["SQ_WAVES"]),len(c["GRBM_GUI_ACTIVE"])))
for tot,nm,gui,v,s,t,vw,n in sorted(rows,reverse=True)[:14]:
    print("%-26s x%-3d cycles/launch %8.0f  VALUBusy %5.1f%%  SALUBusy %5.1f%%  TA busy %5.1f%%  VALU/wave %6.0f"%(nm,n,gui,v,s,t,vw))
PY
