"""Summarize the counter passes of tools/pmc_density.sh (gpurun_out/pmc_<tag>/{sq1,sq2,tcp1,tcc1}/p_counter_collection.csv)
into one JSON: per gather kernel the mean counter value per dispatch plus a few derived figures.
usage: python tools/summarize_pmc.py gpurun_out/pmc_g profiles/r01_g_gather_pmc_2M.json "<workload text>" """
import csv
import json
import os
import sys
from collections import defaultdict

KERNELS = [("k_density_staged", "density_staged"), ("k_forces_fast", "forces_fast"), ("k_density_tiled", "density_tiled"), ("k_forces_lists", "forces_lists"), ("k_forces_tiled", "forces_tiled"),
           ("k_density_ref", "density_reference_order"), ("k_forces_ref", "forces_reference_order")]


def main():
    src, dst, workload = sys.argv[1], sys.argv[2], sys.argv[3]
    acc = defaultdict(lambda: defaultdict(list))
    for sub in sorted(os.listdir(src)):
        path = os.path.join(src, sub, "p_counter_collection.csv")
        if not os.path.isfile(path):
            continue
        for r in csv.DictReader(open(path)):
            for key, name in KERNELS:
                if key in r["Kernel_Name"]:
                    acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
                    break
    out = {"workload": workload,
           "note": "rocprofv3 --pmc, one pass per counter group; values averaged over dispatches; SQ_* are summed over the chip",
           "kernels": {}}
    for name, ctrs in acc.items():
        k = {c: sum(v) / len(v) for c, v in ctrs.items()}
        waves = k.get("SQ_WAVES", 0)
        if waves:
            for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_LDS"):
                if c in k:
                    k["derived_per_wave_" + c] = k[c] / waves
        if k.get("SQ_INSTS_VALU") and k.get("SQ_THREAD_CYCLES_VALU"):
            k["derived_avg_active_lanes_per_VALU_inst"] = k["SQ_THREAD_CYCLES_VALU"] / k["SQ_INSTS_VALU"] / 4.0
        if k.get("SQ_LDS_IDX_ACTIVE"):
            k["derived_LDS_bank_conflict_fraction"] = k.get("SQ_LDS_BANK_CONFLICT", 0) / k["SQ_LDS_IDX_ACTIVE"]
        if k.get("TCC_REQ_sum"):
            k["derived_L2_hit_rate"] = k.get("TCC_HIT_sum", 0) / k["TCC_REQ_sum"]
        if k.get("SQ_WAVE_CYCLES") and k.get("SQ_WAIT_ANY"):
            k["derived_fraction_of_wave_cycles_waiting"] = k["SQ_WAIT_ANY"] / k["SQ_WAVE_CYCLES"]
        if k.get("SQ_WAVE_CYCLES") and k.get("SQ_ACTIVE_INST_VALU"):
            k["derived_fraction_of_wave_cycles_issuing_VALU"] = k["SQ_ACTIVE_INST_VALU"] / k["SQ_WAVE_CYCLES"]
        out["kernels"][name] = k
    json.dump(out, open(dst, "w"), indent=1)
    for name, k in out["kernels"].items():
        print(name, {c: round(v, 3) for c, v in k.items() if c.startswith("derived")})


if __name__ == "__main__":
    main()
