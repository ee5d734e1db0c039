#!/bin/bash
# Busy / memory counters of the density and force launches ON THE BENCH'S OWN STATE (NS scene, 10 M particles, after 3000 steps at
# dt = 2.5e-4 s) for the in-tree library and the tools/_bin variants named on the command line: two rocprofv3 --pmc passes each.
# BUSY_MODE=partial (the first version of this tool): over `density_ablate2.py time <state>`, 5 evaluations of the density + force stages of
# a PARTIAL step (the unfused force launch); default: over the bench's own command, full (fused) steps, the last 5 dispatches of each kernel.  Writes gpurun_out/busyflow/busy.json (raw per-dispatch
# averages and derived fractions, the per-XCD correction as in tools/busy_counters.sh); raw profiler output stays in /tmp on the box.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=/tmp/busyflow; mkdir -p $OUT $R/gpurun_out/busyflow
CMD="$R/bench.py --steps 5 --warmup 20 --resting-steps 0 --no-cpu-baseline"
if [ "$BUSY_MODE" = partial ]; then CMD="$R/tools/density_ablate2.py time /tmp/flow3000.npz"; fi
[ "$BUSY_MODE" != partial ] || [ -f /tmp/flow3000.npz ] || NEREUS_ABL_DT=2.5e-4 timeout -k 10 300 python3 $R/tools/density_ablate2.py save 3000 /tmp/flow3000.npz > $OUT/save.log 2>&1 || { tail -5 $OUT/save.log; exit 1; }
for v in main "$@"; do
  if [ $v = main ]; then unset NEREUS_HIP_LIB; else export NEREUS_HIP_LIB=$R/tools/_bin/libnereus_hip_$v.so; fi
  timeout -k 10 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES TA_BUSY_avr SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY -d $OUT/${v}_a -o p --output-format csv -- python3 $CMD > $OUT/${v}_a.log 2>&1 || { tail -5 $OUT/${v}_a.log; exit 1; }
  timeout -k 10 300 rocprofv3 --pmc GRBM_GUI_ACTIVE TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCC_HIT_sum TCC_MISS_sum -d $OUT/${v}_b -o p --output-format csv -- python3 $CMD > $OUT/${v}_b.log 2>&1 || { tail -5 $OUT/${v}_b.log; exit 1; }
done
python3 - "$OUT" main "$@" <<'PY'
import csv,sys,glob,collections,json,os
out=sys.argv[1]
doc={"workload":"SESPH dam-break NS scene (10,077,696 particles + tank), fp32, Muller kernels, state after 3000 steps at dt = 2.5e-4 s (the bench's "
     "timed window); rocprofv3 --pmc, two passes; values = averages per dispatch",
     "correction":"GRBM_GUI_ACTIVE is reported summed over the 8 XCDs: cycles = GRBM_GUI_ACTIVE / 8 (see tools/busy_counters.sh)","variants":{}}
for v in sys.argv[2:]:
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for p in "ab":
        f=glob.glob("%s/%s_%s/**/*counter_collection.csv"%(out,v,p),recursive=True)[0]
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"]
            nm="density" if "k_density_tiled" in k else ("forces" if "k_forces_lists" in k else None)
            if nm: acc[nm][r["Counter_Name"]+("" if p=="a" or r["Counter_Name"]!="GRBM_GUI_ACTIVE" else "_b")].append(float(r["Counter_Value"]))
    for nm in acc:
        for k in acc[nm]: acc[nm][k]=acc[nm][k][-5:]   # the last five dispatches: the timed window of the bench's command
    doc["variants"][v]={}
    for nm,c in acc.items():
        m={k:sum(x)/len(x) for k,x in c.items()}
        gui=m["GRBM_GUI_ACTIVE"]; guib=m.get("GRBM_GUI_ACTIVE_b",gui)
        d={"cycles":gui/8,"VALUBusy":8*m["SQ_ACTIVE_INST_VALU"]/256/gui,"TA_busy":8*m["TA_BUSY_avr"]/gui,
           "VALU_per_wave":m["SQ_INSTS_VALU"]/m["SQ_WAVES"],"VMEM_RD_per_wave":m["SQ_INSTS_VMEM_RD"]/m["SQ_WAVES"],
           "occupancy":4*m["SQ_WAVE_CYCLES"]/(gui/8)/256/32,"wave_cycles_waiting":m["SQ_WAIT_INST_ANY"]/m["SQ_WAVE_CYCLES"],
           "L1_accesses_per_wave":m["TCP_TOTAL_CACHE_ACCESSES_sum"]/m["SQ_WAVES"],"L1_to_L2_read_req_per_wave":m["TCP_TCC_READ_REQ_sum"]/m["SQ_WAVES"],
           "L1_hit_rate":1-m["TCP_TCC_READ_REQ_sum"]/m["TCP_TOTAL_CACHE_ACCESSES_sum"],"L2_hit_rate":m["TCC_HIT_sum"]/(m["TCC_HIT_sum"]+m["TCC_MISS_sum"]),
           "L1_accesses_per_cycle_per_CU":m["TCP_TOTAL_CACHE_ACCESSES_sum"]/256/(guib/8),
           "TCP_pending_stall_fraction":m["TCP_PENDING_STALL_CYCLES_sum"]/256/(guib/8)}
        doc["variants"][v][nm]={"raw":m,"dispatches":len(next(iter(c.values()))),"derived":d}
        print("%-10s %-8s"%(v,nm)+"  ".join("%s %.3g"%(k,x) for k,x in d.items()))
json.dump(doc,open(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/busyflow/busy.json","w"),indent=1)
PY
