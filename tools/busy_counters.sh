#!/bin/bash
# VALU / SALU / TA busy of the density and force kernels for library variants (2.1 M-particle scene at rest, one PMC pass each)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc2; mkdir -p $OUT
export NEREUS_ABLATE_NOREF=1
for v in main "$@"; do
  if [ $v = main ]; then unset NEREUS_HIP_LIB; else export NEREUS_HIP_LIB=$R/tools/_bin/libnereus_hip_$v.so; fi
  rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INST_CYCLES_SALU SQ_INSTS_SALU SQ_WAVES TA_BUSY_avr SQ_WAVE_CYCLES -d $OUT/$v -o p --output-format csv -- python3 $R/tools/ablate_density.py 128,128,128 > $OUT/$v.log 2>&1 || { tail -5 $OUT/$v.log; exit 1; }
done
python3 - "$OUT" main "$@" <<'PY'
import csv,sys,glob,collections
out=sys.argv[1]
for v in sys.argv[2:]:
    f=glob.glob("%s/%s/**/*counter_collection.csv"%(out,v),recursive=True)[0]
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        nm="density" if "k_density_tiled" in k else ("forces" if "k_forces_lists" in k else None)
        if nm: acc[nm][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for nm,c in acc.items():
        m={k:sum(x)/len(x) for k,x in c.items()}
        gui=m["GRBM_GUI_ACTIVE"]
        print("%-12s %-8s GUI %8.0f cyc  VALUBusy %5.1f%%  SALUBusy %5.1f%%  TA_BUSY/GUI %5.1f%%  VALU/wave %6.0f  SALU/wave %5.0f  occupancy %4.1f%%" % (v,nm,gui,100*m["SQ_ACTIVE_INST_VALU"]/256/gui,100*m["SQ_INST_CYCLES_SALU"]/256/gui,100*m["TA_BUSY_avr"]/gui,m["SQ_INSTS_VALU"]/m["SQ_WAVES"],m["SQ_INSTS_SALU"]/m["SQ_WAVES"],400*m["SQ_WAVE_CYCLES"]/gui/256/32))
PY
