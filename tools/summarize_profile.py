"""Turn the raw output of tools/profile_bench.sh (gpurun_out/prof_<tag>/) into the committed summaries:
profiles/<name>_kernel_stats.csv (copy of rocprofv3's --stats table) and profiles/<name>_hbm_traffic.json
(HBM bytes per launch per kernel from the FETCH_SIZE / WRITE_SIZE passes, raw and with the gfx950 correction of
MI355X_MICROARCH.md: 2*FETCH_SIZE + WRITE_SIZE; counter values are KB per dispatch).
usage: python tools/summarize_profile.py gpurun_out/prof_r01e r01_e_ns10M "workload text" particles"""
import csv
import json
import os
import shutil
import sys
from collections import defaultdict

SHORT = [("k_displacement_lists", "k_displacement_lists"), ("k_advection_lists", "k_advection_lists"), ("k_sumdij_lists", "k_sumdij_lists"),
         ("k_pressure_lists", "k_pressure_lists"), ("k_pforce_lists", "k_pforce_lists"), ("k_iisph_integrate", "k_iisph_integrate"),
         ("k_sum_partial", "k_sum_partial"), ("k_sum_final", "k_sum_final"),
         ("k_density_staged", "k_density_staged"), ("k_forces_fast", "k_forces_fast"), ("k_wall_count", "k_wall_count"),
         ("k_wall_compact", "k_wall_compact"), ("k_density_tiled", "k_density_tiled"), ("k_forces_lists", "k_forces_lists"), ("k_forces_tiled", "k_forces_tiled"),
         ("k_reorder_merged", "k_reorder_merged"), ("k_reorder_boundary", "k_reorder_boundary"), ("k_reorder", "k_reorder"),
         ("k_clear_cells", "k_clear_cells"), ("k_hash", "k_hash"), ("k_resort_split", "k_resort_split"),
         ("k_resort_scan_tiles", "k_resort_scan_tiles"), ("k_resort_count", "k_resort_count"), ("k_integrate", "k_integrate"),
         ("onesweep_histograms", "radix_sort_histograms"), ("onesweep_scan", "radix_sort_scan_histograms"),
         ("radix_sort_onesweep", "radix_sort_onesweep"), ("merge", "rocprim_merge"), ("fillBuffer", "fillBuffer"),
         ("copyBuffer", "copyBuffer")]


def short(name):
    for key, s in SHORT:
        if key in name:
            return s
    return name[:60]


def per_kernel(path, counter, last=0):
    """counter values per kernel, in dispatch order; last > 0 keeps only the last `last` dispatches of each kernel"""
    acc = defaultdict(list)
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r.get("Dispatch_Id", 0)))
    for r in rows:
        acc[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    if last:
        acc = {k: v[-last:] for k, v in acc.items()}
    return acc


def tail_stats(trace_csv, dst_csv, steps):
    """per-kernel launch statistics over the LAST `steps` steps of a kernel trace (a step = one k_forces_* / k_iisph_integrate launch)"""
    rows = list(csv.DictReader(open(trace_csv)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    marks = [int(r["Start_Timestamp"]) for r in rows if "k_forces_" in r["Kernel_Name"] or "k_iisph_integrate" in r["Kernel_Name"]]
    t0 = marks[-steps] if len(marks) >= steps else marks[0]
    first_stage = min((int(r["Start_Timestamp"]) for r in rows if int(r["Start_Timestamp"]) >= t0), default=t0)
    acc = defaultdict(list)
    for r in rows:
        if int(r["Start_Timestamp"]) >= first_stage:
            acc[short(r["Kernel_Name"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    total = sum(sum(v) for v in acc.values())
    with open(dst_csv, "w") as f:
        f.write('"Name","Calls","TotalDurationNs","AverageNs","Percentage","CallsPerStep"\n')
        for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
            f.write('"%s",%d,%d,%.1f,%.2f,%.2f\n' % (k, len(v), sum(v), sum(v) / len(v), 100.0 * sum(v) / total, len(v) / float(steps)))
    return acc


def main():
    """usage: summarize_profile.py <prof dir> <name> "<workload>" <particles> [tail steps]
    with `tail steps` the statistics cover only the last so-many steps of the traced run (developed-flow profiles)"""
    src, name, workload, particles = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
    tail = int(sys.argv[5]) if len(sys.argv) > 5 else 0
    root = os.environ.get("NEREUS_PROFILE_OUT") or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
    os.makedirs(root, exist_ok=True)
    if tail:
        tail_stats(os.path.join(src, "trace", "t_kernel_trace.csv"), os.path.join(root, name + "_kernel_stats.csv"), tail)
    else:
        shutil.copy(os.path.join(src, "trace", "t_kernel_stats.csv"), os.path.join(root, name + "_kernel_stats.csv"))
    f = per_kernel(os.path.join(src, "fetch", "f_counter_collection.csv"), "FETCH_SIZE", 5 if tail else 0)
    w = per_kernel(os.path.join(src, "write", "w_counter_collection.csv"), "WRITE_SIZE", 5 if tail else 0)
    out = {"workload": workload,
           "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes; counter values are KB per dispatch; "
                   "corrected = 2*FETCH_SIZE + WRITE_SIZE as MI355X_MICROARCH.md prescribes for gfx950 "
                   "(uncalibrated for gather access patterns)",
           "particles": particles, "kernels": {}}
    for k in sorted(set(f) | set(w)):
        fk, wk = f.get(k, [0.0]), w.get(k, [0.0])
        fa, wa = sum(fk) / len(fk), sum(wk) / len(wk)
        out["kernels"][k] = {"FETCH_SIZE_KB_avg": fa, "FETCH_SIZE_dispatches": len(fk), "WRITE_SIZE_KB_avg": wa,
                             "WRITE_SIZE_dispatches": len(wk), "hbm_bytes_per_launch_corrected": 1024.0 * (2 * fa + wa),
                             "hbm_bytes_per_launch_raw": 1024.0 * (fa + wa)}
    json.dump(out, open(os.path.join(root, name + "_hbm_traffic.json"), "w"), indent=1)
    for k, v in out["kernels"].items():
        print("%-28s %4d launches  %8.1f MB/launch (corrected)" % (k, v["FETCH_SIZE_dispatches"], v["hbm_bytes_per_launch_corrected"] / 1e6))


if __name__ == "__main__":
    main()
