#!/bin/bash
# busy counters of the IISPH chain's kernels at config C3 (one rocprofv3 --pmc pass over the bench line, last dispatches of each kernel)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=/tmp/busyc3; mkdir -p $OUT $R/gpurun_out/busyc3
timeout -k 10 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES TA_BUSY_avr SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY -d $OUT/a -o p --output-format csv -- python3 $R/bench.py --solver iisph --config C3 --steps 5 --warmup 5 --no-cpu-baseline > $OUT/a.log 2>&1 || { tail -5 $OUT/a.log; exit 1; }
timeout -k 10 300 rocprofv3 --pmc GRBM_GUI_ACTIVE TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCC_HIT_sum TCC_MISS_sum -d $OUT/b -o p --output-format csv -- python3 $R/bench.py --solver iisph --config C3 --steps 5 --warmup 5 --no-cpu-baseline > $OUT/b.log 2>&1 || { tail -5 $OUT/b.log; exit 1; }
python3 - "$OUT" <<'PY'
import csv,sys,glob,collections,json,os
out=sys.argv[1]
names=["k_pressure_lists","k_sumdij_lists","k_displacement_lists","k_advection_lists","k_pforce_lists","k_density_tiled"]
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for p in "ab":
    f=glob.glob("%s/%s/**/*counter_collection.csv"%(out,p),recursive=True)[0]
    for r in csv.DictReader(open(f)):
        for nm in names:
            if nm in r["Kernel_Name"]:
                acc[nm][r["Counter_Name"]+("" if p=="a" or r["Counter_Name"]!="GRBM_GUI_ACTIVE" else "_b")].append(float(r["Counter_Value"]))
doc={"workload":"config C3 (IISPH, 4,096,000 particles), bench.py --solver iisph --config C3, rocprofv3 --pmc, two passes, last 5 dispatches per kernel","kernels":{}}
for nm,c in acc.items():
    m={k:sum(x[-5:])/len(x[-5:]) for k,x in c.items()}
    gui=m["GRBM_GUI_ACTIVE"]; guib=m.get("GRBM_GUI_ACTIVE_b",gui)
    d={"cycles":gui/8,"VALUBusy":8*m["SQ_ACTIVE_INST_VALU"]/256/gui,"TA_busy":8*m["TA_BUSY_avr"]/gui,"VALU_per_wave":m["SQ_INSTS_VALU"]/m["SQ_WAVES"],
       "VMEM_RD_per_wave":m["SQ_INSTS_VMEM_RD"]/m["SQ_WAVES"],"occupancy":4*m["SQ_WAVE_CYCLES"]/(gui/8)/256/32,"wave_cycles_waiting":m["SQ_WAIT_INST_ANY"]/m["SQ_WAVE_CYCLES"],
       "L1_accesses_per_wave":m["TCP_TOTAL_CACHE_ACCESSES_sum"]/m["SQ_WAVES"],"L1_hit_rate":1-m["TCP_TCC_READ_REQ_sum"]/m["TCP_TOTAL_CACHE_ACCESSES_sum"],
       "L2_hit_rate":m["TCC_HIT_sum"]/(m["TCC_HIT_sum"]+m["TCC_MISS_sum"]),"L1_accesses_per_cycle_per_CU":m["TCP_TOTAL_CACHE_ACCESSES_sum"]/256/(guib/8),
       "TCP_pending_stall_fraction":m["TCP_PENDING_STALL_CYCLES_sum"]/256/(guib/8)}
    doc["kernels"][nm]={"raw":m,"derived":d}
    print("%-22s"%nm+"  ".join("%s %.3g"%(k,x) for k,x in d.items()))
json.dump(doc,open(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/busyc3/busy_c3.json","w"),indent=1)
PY
