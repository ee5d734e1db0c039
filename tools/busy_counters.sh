#!/bin/bash
# VALU / SALU / TA busy of the density and force kernels for library variants (2.1 M-particle scene at rest, one PMC pass each).
# Writes gpurun_out/pmc2/busy.json: RAW per-dispatch counter averages + the derived fractions with the correction written out
# (profiles/r03_busy_2M.json is a copy of it).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=/tmp/pmc2; mkdir -p $OUT $R/gpurun_out/pmc2
export NEREUS_ABLATE_NOREF=1
for v in main "$@"; do
  if [ $v = main ]; then unset NEREUS_HIP_LIB; else export NEREUS_HIP_LIB=$R/tools/_bin/libnereus_hip_$v.so; fi
  rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INST_CYCLES_SALU SQ_INSTS_SALU SQ_WAVES TA_BUSY_avr SQ_WAVE_CYCLES -d $OUT/$v -o p --output-format csv -- python3 $R/tools/ablate_density.py 128,128,128 > $OUT/$v.log 2>&1 || { tail -5 $OUT/$v.log; exit 1; }
done
python3 - "$OUT" main "$@" <<'PY'
import csv,sys,glob,collections,json
out=sys.argv[1]
doc={"workload":"SESPH dam-break 128^3 = 2,097,152 particles + tank, fp32, Muller kernels, resting column; rocprofv3 --pmc (one pass), "
     "values = averages per dispatch of the kernel",
     "correction":"rocprofv3 reports GRBM_GUI_ACTIVE summed over the 8 XCDs of the MI355X while a kernel's wall time in cycles is the per-XCD "
     "value: cycles = GRBM_GUI_ACTIVE / 8.  SQ_ACTIVE_INST_VALU and SQ_INST_CYCLES_SALU are summed over all SIMDs / CUs (256 CUs x 4 SIMDs) "
     "and counted in quad-cycles (x4); TA_BUSY_avr is an average over the TA instances, in the summed-over-XCDs cycle base.  So "
     "VALUBusy = 4 * SQ_ACTIVE_INST_VALU / (1024 SIMDs) / cycles, SALUBusy = 4 * SQ_INST_CYCLES_SALU / (256 CUs) / 4 / cycles ... "
     "written out below as the exact expressions that were evaluated",
     "formulas":{"cycles":"GRBM_GUI_ACTIVE / 8","VALUBusy":"SQ_ACTIVE_INST_VALU / 256 / GRBM_GUI_ACTIVE * 8 (= rocprofv3's VALUBusy definition with the per-XCD cycle count)",
                 "SALUBusy":"SQ_INST_CYCLES_SALU / 256 / GRBM_GUI_ACTIVE * 8","TA_busy":"TA_BUSY_avr / GRBM_GUI_ACTIVE * 8",
                 "VALU_per_wave":"SQ_INSTS_VALU / SQ_WAVES","SALU_per_wave":"SQ_INSTS_SALU / SQ_WAVES",
                 "occupancy":"SQ_WAVE_CYCLES * 4 / (GRBM_GUI_ACTIVE / 8) / 256 / 32 (wave slots: 8 per SIMD x 4 SIMDs)"},
     "variants":{}}
for v in sys.argv[2:]:
    f=glob.glob("%s/%s/**/*counter_collection.csv"%(out,v),recursive=True)[0]
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        nm="density" if "k_density_tiled" in k else ("forces" if "k_forces_lists" in k else None)
        if nm: acc[nm][r["Counter_Name"]].append(float(r["Counter_Value"]))
    doc["variants"][v]={}
    for nm,c in acc.items():
        m={k:sum(x)/len(x) for k,x in c.items()}
        gui=m["GRBM_GUI_ACTIVE"]
        d={"raw":m,"dispatches":len(next(iter(c.values()))),"derived":{
            "cycles":gui/8,"VALUBusy":8*m["SQ_ACTIVE_INST_VALU"]/256/gui,"SALUBusy":8*m["SQ_INST_CYCLES_SALU"]/256/gui,
            "TA_busy":8*m["TA_BUSY_avr"]/gui,"VALU_per_wave":m["SQ_INSTS_VALU"]/m["SQ_WAVES"],"SALU_per_wave":m["SQ_INSTS_SALU"]/m["SQ_WAVES"],
            "occupancy":4*m["SQ_WAVE_CYCLES"]/(gui/8)/256/32}}
        doc["variants"][v][nm]=d
        print("%-12s %-8s cycles %8.0f  VALUBusy %5.1f%%  SALUBusy %5.1f%%  TA busy %5.1f%%  VALU/wave %6.0f  SALU/wave %5.0f  occupancy %4.1f%%" % (
            v,nm,gui/8,100*d["derived"]["VALUBusy"],100*d["derived"]["SALUBusy"],100*d["derived"]["TA_busy"],d["derived"]["VALU_per_wave"],d["derived"]["SALU_per_wave"],100*d["derived"]["occupancy"]))
import os
json.dump(doc,open(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/pmc2/busy.json","w"),indent=1)
PY
