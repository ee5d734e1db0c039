#!/usr/bin/env python3
"""bench.py — SESPH dam-break throughput on MI355X (BASELINE.json metric: particle-steps/s + ms/step).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--config NS|C2|C4|...]

A "step" is one update() of the solver (hash → radix sort → cell ranges + reorder → density/pressure →
forces → integrate) over every particle of the scene, with the particle state already resident in HBM when
the timed region starts.  N=1 default workload: the north-star size, a 216^3 = 10,077,696-particle SESPH
dam-break (fp32, Muller kernels) in its tank of boundary particles.  For N>1 (launched by torchrun, one rank
per GPU) the block is N times longer in x and slab-partitioned (weak scaling), see nereus_amd/slab.py.

What is timed is the BROKEN dam, not the resting column (round 3): before the contract's warm-up the run does an
untimed spin-up (--spin-up, default 3000 steps) under a time step that respects the CFL limit throughout
(--dt, default 2.5e-4 s: the reference's fixed 1e-3 s passes the limit of this 8.8 m column around step 250 and
blows up before step 1000).  `config.spin_up_steps`, `config.dt` and `cfl_ok` say so in the
line; the resting column (steps 20-120 of the same run) is the `resting` sub-record.

Rank 0 prints ONE JSON line: the driver contract plus
  roofline     dominant kernel, ALGORITHMIC bytes per launch / its mean HIP-event duration in the timed
               region, against the 8 TB/s HBM3E peak (MI355X_MICROARCH.md)
  cpu_baseline the CPU oracle (a port, OpenMP) on a bounded sample of the same workload, host cores of
               this box — a reported baseline, not the optimisation target.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # spec, /opt/skills/guides/MI355X_MICROARCH.md (6290 measured-achievable)
TRAFFIC_PROFILE = "r03b_ns10M_flowing_hbm_traffic.json"  # committed rocprofv3 PMC summary (of the same window) the `traffic` field is read from
DEFAULT_SPIN_UP, DEFAULT_DT = 3000, 2.5e-4



def stage_bytes(real_bytes=4):
    """algorithmic bytes per particle per launch of each stage (V = vec4, S = scalar, U = 4 B):
    fp32 V=16 S=4, fp64 V=32 S=8"""
    V, S, U = 4 * real_bytes, real_bytes, 4
    return {"hash": V + 2 * U, "reorder": 2 * U + 4 * V + 2 * S, "density": V + 2 * S, "forces": 3 * V + 2 * S,
            "integrate": 5 * V}


def iisph_stage_bytes(real_bytes=4):
    """IISPH stages: density V+S, displacement 6V+S, advection 5V+5S, per solver iteration sumDij 2V+2S +
    pressure 4V+7S + reduce S (times L iterations per step), pressure force 2V+2S, integrate 5V"""
    V, S = 4 * real_bytes, real_bytes
    return {"i_density": V + S, "i_displacement": 6 * V + S, "i_advection": 5 * V + 5 * S, "i_solve": 6 * V + 10 * S,
            "i_pforce": 2 * V + 2 * S, "i_integrate": 5 * V}


def fused_forces_bytes(real_bytes=4):
    """a full step on the production kernels runs forces + integrate + next-step hash as ONE launch: its algorithmic
    bytes are the sum of the three reference stages it implements"""
    b = stage_bytes(real_bytes)
    return b["forces"] + b["integrate"] + b["hash"]


STAGE_BYTES_F32 = stage_bytes(4)
IISPH_STAGE_BYTES_F32 = iisph_stage_bytes(4)
FUSED_FORCES_BYTES_F32 = fused_forces_bytes(4)
KERNEL_OF_STAGE = {"forces": "k_forces_lists", "density": "k_density_tiled", "reorder": "k_reorder_merged", "hash": "k_hash",
                   "integrate": "k_integrate", "sort": "k_resort_split"}
TRAFFIC_PROFILE_KERNELS = KERNEL_OF_STAGE


def per_stage_roofline(warm, n, num_cells, production, real_bytes=4):
    out = {}
    sb = stage_bytes(real_bytes)
    fused = production and "integrate" not in warm
    bits = max(1, int(np.ceil(np.log2(max(2, num_cells)))))
    for name, (ms, launches) in warm.items():
        if not launches or ms <= 0:
            continue
        if name == "sort":
            bpp = 16 * ((bits + 7) // 8) + 4
        elif name == "forces" and fused:
            bpp = fused_forces_bytes(real_bytes)
        else:
            bpp = sb.get(name)
        if bpp is None:
            continue
        gbs = bpp * n / (ms / launches * 1e-3) / 1e9
        out[name] = {"ms": ms / launches, "algorithmic_bytes_per_particle": bpp, "achieved_GBs": gbs, "frac": gbs / HBM_PEAK_GBS}
    return out


def measured_traffic(stage, n):
    """HBM bytes per launch of the stage's kernel from the committed rocprofv3 PMC profile (2*FETCH_SIZE + WRITE_SIZE,
    the gfx950 correction of MI355X_MICROARCH.md), scaled by particle count; None if no profile is committed."""
    path = os.path.join(ROOT, "profiles", TRAFFIC_PROFILE)
    try:
        doc = json.load(open(path))
        k = doc["kernels"][KERNEL_OF_STAGE[stage]]
        return k["hbm_bytes_per_launch_corrected"] * (n / float(doc["particles"]))
    except Exception:
        return None


def sesph_bytes_per_particle_step(num_cells, real_bytes=4):
    """B = 268 + 16 P (fp32) / 516 + 16 P (fp64), P = radix passes of 8 bits over log2(numCells) key bits."""
    bits = max(1, int(np.ceil(np.log2(max(2, num_cells)))))
    passes = (bits + 7) // 8
    base = 268 if real_bytes == 4 else 516
    return base + 16 * passes, passes


def usable_cores():
    """CPU threads this process may really use: affinity mask capped by the cgroup CPU quota (the GPU box
    exposes 256 logical CPUs in the mask but grants a 16-CPU share)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0:
            n = min(n, max(1, q // per))
    except Exception:
        pass
    return int(os.environ.get("NEREUS_CPU_THREADS", min(n, 16)))


def cpu_baseline(seconds_target=15.0):
    """Time the CPU oracle (restatement of the reference algorithm, OpenMP over particles) on the C1 scene:
    the same generator and parameters as the GPU workload at 32^3 = 32,768 particles."""
    from nereus_amd import scene
    from tests.oracle_lib import SESPH, Oracle

    cores = usable_cores()
    p = Oracle.default_params(SESPH)
    sc = scene.dam_break("C1", h=float(p["interactionRadius"][0]), kpoly=float(p["kpoly"][0]))
    o = Oracle(p, solver=SESPH, threads=cores)
    o.set_particles(sc["pos"], sc["vel"])
    o.set_boundaries(sc["bi"], sc["vbi"], update_grid=True)
    o.step(2)  # warm-up
    n = len(sc["pos"])
    steps = 0
    t0 = time.perf_counter()
    while True:
        o.step(5)
        steps += 5
        dt = time.perf_counter() - t0
        if dt >= seconds_target or steps >= 2000:
            break
    return {
        "value": n * steps / dt,
        "unit": "particle-steps/s",
        "cores": cores,
        "kind": "port",
        "sample": "C1: 32^3=32768-particle SESPH dam-break with tank boundaries, %d steps, OpenMP oracle" % steps,
        "ms_per_step": 1e3 * dt / steps,
    }


def timed_window(s, steps, torch):
    """`steps` steps bracketed by synchronisations; returns seconds"""
    s.synchronize()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    s.step(steps)
    s.synchronize()
    torch.cuda.synchronize()
    return time.perf_counter() - t0


def flow_state(s, capi, n):
    """max |v| and the CFL limit of the state (the reference's rule, sph.cpp:217-231: dt <= 0.4 h / (c_s + |v|max)), plus the
    hit-list diagnostics of the last step"""
    P = s.params
    h, dt, cs = float(P["interactionRadius"][0]), float(P["timestep"][0]), float(P["soundSpeed"][0])
    vmax = s.max_velocity()
    rec = {"vmax": vmax, "cfl_dt_limit": 0.4 * h / (cs + vmax), "dt": dt, "cells_per_step": vmax * dt / h}
    try:
        rec["mover_fraction_last_step"] = s.get_stat(capi.STAT_MOVERS) / n
        rec["hit_list_overflow_fraction"] = s.get_stat(capi.STAT_HIT_OVERFLOW) / n
        rec["neighbours_mean"] = s.get_stat(capi.STAT_HIT_MEAN)
        rec["neighbours_max"] = s.get_stat(capi.STAT_HIT_MAX)
    except capi.NereusError as e:  # (reference-order kernels keep no hit lists)
        rec["stats_unavailable"] = str(e)
    return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="NS", help="scene lattice: NS (216^3), C2 (100^3), C4, C1 ... or nx,ny,nz")
    ap.add_argument("--solver", default="sesph", choices=["sesph", "iisph"],
                    help="sesph = the BASELINE metric (default); iisph = config 3 style run of the IISPH chain (N=1 only)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="N>1: weak = every rank gets a block of --config size (default, the driver contract); strong = the "
                         "--config lattice is split across the ranks in x (e.g. --config C4 --gpus 8: 16M particles, 2M per GPU)")
    ap.add_argument("--precision", type=int, default=32, choices=[32, 64],
                    help="SReal: 32 (default, the reference's shipped build) or 64 (DOUBLE_PRECISION=1, config 5)")
    ap.add_argument("--kernel-set", default="muller", choices=["muller", "monaghan"],
                    help="KERNEL_SET: muller (default) or monaghan (config 5)")
    ap.add_argument("--arith", default="exact", choices=["exact", "fast"],
                    help="exact = every float sum in the reference's order with IEEE arithmetic (bit-identical to the CPU oracle); "
                         "fast = NRS_FLAG_FAST_ARITH (tolerance mode, fp32 Muller SESPH; indices stay bit-exact)")
    ap.add_argument("--spin-up", type=int, default=None, metavar="STEPS",
                    help="untimed steps BEFORE the contract's warm-up that take the scene from the resting column to the broken dam "
                         "(default %d for SESPH, 0 for IISPH; 0 = time the resting column as rounds 1-2 did)" % DEFAULT_SPIN_UP)
    ap.add_argument("--dt", type=float, default=None,
                    help="fixed time step in seconds (default %g for SESPH: CFL-stable through the spin-up and the timed region; "
                         "IISPH keeps its constructor default).  The reference's own 1e-3 s (sph.cpp:60): --dt 1e-3" % DEFAULT_DT)
    ap.add_argument("--developed", type=int, default=None, help=argparse.SUPPRESS)  # (rounds 1-2: window behind the timed region; ignored)
    ap.add_argument("--developed-steps", type=int, default=None, help=argparse.SUPPRESS)
    ap.add_argument("--resting-steps", type=int, default=100, help="length of the `resting` sub-record's window (0 = off)")
    ap.add_argument("--iisph-max-iters", type=int, default=2,
                    help="N>1 IISPH: solver iterations per step the halo is sized for (halo = 2 * iterations + 4 cells)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--reference-order", action="store_true", help="bench the reference-order kernels instead")
    ap.add_argument("--full-sort", action="store_true", help="sort all pairs from scratch every step (NRS_FLAG_FULL_SORT)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs torchrun: python -m torch.distributed.run --nnodes=1 --nproc-per-node %d "
                             "--master-addr 127.0.0.1 --master-port P bench.py --gpus %d ..." % (args.gpus, args.gpus, args.gpus))
        raise SystemExit("--gpus (%d) != WORLD_SIZE (%d)" % (args.gpus, world))

    import torch

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    if os.environ.get("NEREUS_BENCH_BACKEND") == "gloo" and local_rank >= torch.cuda.device_count():
        local_rank %= torch.cuda.device_count()  # rehearsal of the N-rank path on a box with fewer GPUs (host-staged exchange only)
    torch.cuda.set_device(local_rank)

    from nereus_amd import capi, scene

    lattice = scene.CONFIGS[args.config] if args.config in scene.CONFIGS else tuple(int(v) for v in args.config.split(","))

    if world > 1 or os.environ.get("NEREUS_BENCH_FORCE_SLAB"):
        from nereus_amd import slab
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")

        if args.scaling == "strong":
            if lattice[0] % world:
                raise SystemExit("--scaling strong needs the x extent of the lattice (%d) divisible by --gpus" % lattice[0])
            lattice = (lattice[0] // world,) + tuple(lattice[1:])
        # RCCL prints a version banner on STDOUT when its communicator comes up: keep fd 1 for the ONE JSON line
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            result = slab.bench_main(args, lattice, rank, world, local_rank)
        finally:
            sys.stdout.flush()
            os.dup2(saved, 1)
            os.close(saved)
        result["scaling"] = args.scaling
        if rank == 0:
            print(json.dumps(result))
        return

    # ---------------------------------------------------------------- single GPU
    from nereus_amd.params import default_params

    iisph = args.solver == "iisph"
    double = args.precision == 64
    real, real_bytes = (np.float64, 8) if double else (np.float32, 4)
    kset = capi.MULLER if args.kernel_set == "muller" else capi.MONAGHAN
    p = default_params(1 if iisph else 0, double)  # constructor defaults (sph/sph.cpp:29-93, iisph/iisph.cpp:28-87)
    t_gen = time.perf_counter()
    sc = scene.dam_break(lattice, h=float(p["interactionRadius"][0]), kpoly=float(p["kpoly"][0]), real=real)
    t_gen = time.perf_counter() - t_gen
    n = len(sc["pos"])
    stream = torch.cuda.current_stream().cuda_stream
    s = capi.Solver(p, n, solver=capi.IISPH if iisph else capi.SESPH, double=double, kernel_set=kset, device=local_rank,
                    stream=stream, reference_order=args.reference_order,
                    flags=(capi.FLAG_FULL_SORT if args.full_sort else 0) | (capi.FLAG_FAST_ARITH if args.arith == "fast" else 0))
    s.set_particles(sc["pos"], sc["vel"])
    s.set_boundaries(sc["bi"], sc["vbi"], update_grid=True)
    P = s.params
    num_cells = int(P["numCells"][0])
    if args.dt is not None or not iisph:   # (set after the grid exists; a new time step keeps every prepared key, nrs_set_params)
        P["timestep"][0] = args.dt if args.dt is not None else DEFAULT_DT
        s.set_params(P)
    spin_up = args.spin_up if args.spin_up is not None else (0 if iisph else DEFAULT_SPIN_UP)
    bpp, passes = sesph_bytes_per_particle_step(num_cells, real_bytes)

    # untimed spin-up to the flowing state; on its way the resting column's window (steps 20-120) as a sub-record
    done = 0
    resting = None
    if spin_up > 0:
        s.step(1)  # (the very first step also pays one-off costs: rocPRIM set-up, the full sort after an upload)
        done = 1
        if not iisph and args.resting_steps > 0 and spin_up >= 20 + args.resting_steps:
            s.step(19)
            s.set_profiling(True)
            t_rest = timed_window(s, args.resting_steps, torch)
            st = {k: v[0] / max(1, v[1]) for k, v in s.stage_ms().items()}
            s.set_profiling(False)
            done = 20 + args.resting_steps
            resting = {"first_step": 20, "steps": args.resting_steps, "ms_per_step": 1e3 * t_rest / args.resting_steps,
                       "value": n * args.resting_steps / t_rest,
                       "whole_step_frac": bpp * n * args.resting_steps / t_rest / 1e9 / HBM_PEAK_GBS, "stage_ms": st}
            resting.update(flow_state(s, capi, n))
        s.step(spin_up - done)
        done = spin_up

    # (the flow state is read BEFORE the warm-up: its reductions and read-backs leave the device idle for a moment, and a short timed
    # region right behind such a gap measures the clock ramp — 20 steps ran 5 % slower that way)
    before = None if iisph else flow_state(s, capi, n)
    # the contract's warm-up (untimed), with every stage timed once to find the dominant kernel
    first = 1 if (args.warmup > 1 and done == 0) else 0
    tail = 2 if args.warmup - first >= 4 else 0   # the last warm-up steps run AFTER the stage times have been read back, so that nothing but
    s.step(first)                                 # the contract's synchronisation separates the warm-up from the timed region (reading ~25
    s.set_profiling(True)                         # event pairs leaves the device idle for about a millisecond; a 20-step region right behind
    s.step(args.warmup - first - tail)            # such a gap ran 3-4 % slower)
    s.synchronize()
    warm = s.stage_ms()
    dominant = max(warm, key=lambda k: warm[k][0]) if warm else "forces"
    dom_id = {v: k for k, v in capi.STAGE_NAMES.items()}[dominant]
    s.set_profiling([dom_id])
    if tail:
        s.step(tail)
        s.synchronize()
        s.set_profiling([dom_id])   # (drops the two warm-up launches from the dominant kernel's statistics)

    dt = timed_window(s, args.steps, torch)
    timed = s.stage_ms()
    dom_ms, dom_launches = timed.get(dominant, (0.0, 0))
    after = None if iisph else flow_state(s, capi, n)

    # sanity: the state is finite
    gp, gv = s.download()
    if not (np.isfinite(gp).all() and np.isfinite(gv).all()):
        raise SystemExit("non-finite state after the run")

    ms_per_step = 1e3 * dt / args.steps
    value = n * args.steps / dt
    isb = iisph_stage_bytes(real_bytes)
    if iisph:  # 448 + 16 P + 136 L bytes per particle-step (fp32), L = solver iterations of the last step
        sb = stage_bytes(real_bytes)
        bpp = (sb["hash"] + 4 + sb["reorder"] + isb["i_density"] + isb["i_displacement"] + isb["i_advection"] + isb["i_pforce"]
               + isb["i_integrate"] + 16 * passes + isb["i_solve"] * s.last_iterations)
    fused = dominant == "forces" and not args.reference_order and "integrate" not in warm
    if dominant in isb:
        dom_bpp = isb[dominant] * (s.last_iterations if dominant == "i_solve" else 1)
    else:
        dom_bpp = fused_forces_bytes(real_bytes) if fused else stage_bytes(real_bytes).get(dominant, 16 * passes + 4 if dominant == "sort" else 0)
    dom_bytes = dom_bpp * n
    dom_avg_ms = dom_ms / max(1, dom_launches)
    achieved = (dom_bytes / (dom_avg_ms * 1e-3)) / 1e9 if dom_avg_ms > 0 else 0.0
    out = {
        "metric": "particle-steps/sec, %s dam-break" % ("IISPH" if iisph else "SESPH"),
        "value": value,
        "unit": "particle-steps/s",
        "n_gpus": 1,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64" if double else "f32",
        "data": "synthetic",
        "config": {
            "workload": "%s dam-break %dx%dx%d = %d particles (+%d tank boundary particles), %s, %s kernels, "
                        "grid %dx%dx%d" % (("IISPH" if iisph else "SESPH",) + lattice + (n, len(sc["bi"]))
                                           + ("fp64 (DOUBLE_PRECISION=1)" if double else "fp32",
                                              "Muller" if kset == capi.MULLER else "Monaghan (KERNEL_SET=0)")
                                           + tuple(int(v) for v in P["gridSize"][0])),
            "particles": n,
            "boundary_particles": int(len(sc["bi"])),
            "num_cells": num_cells,
            "steps_per_s": args.steps / dt,
            "spin_up_steps": spin_up,
            "first_timed_step": spin_up + args.warmup,
            "dt": float(s.params["timestep"][0]),
            "simulated_time_at_window_start_s": (spin_up + args.warmup) * float(s.params["timestep"][0]),
            "kernels": "reference-order" if args.reference_order else "tiled",
            "arith": args.arith if (not double and kset == capi.MULLER and not iisph and not args.reference_order) else "exact",
            "sort": dict(zip(("coherent_resort_steps", "fell_back_to_full_sort"), s.resort_stats())),
            "parallelism": "1 GPU",
        },
        "roofline": {
            "bound": "hbm",
            "kernel": "forces+integrate+hash (one fused launch)" if fused else dominant,
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "traffic": measured_traffic(dominant, n),
            "traffic_source": "profiles/%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes), "
                              "scaled by particle count" % TRAFFIC_PROFILE,
            "kernel_avg_ms": dom_avg_ms,
            "kernel_launches": dom_launches,
            "algorithmic_bytes_per_launch": dom_bytes,
            # (not measured in this run) which unit the gather kernels keep busy: see the committed counter summary
            "limiter_note": ("the two gather launches are within a few per cent of each other; the density launch (the step's only neighbour "
                             "search; it also publishes the hit lists and the gather records) is bound by vector-instruction issue, the "
                             "fused force launch by L1 line accesses and their misses, neither by HBM bytes: "
                             "profiles/r03_busy_flowing_10M.json") if dominant in ("density", "forces") else None,
            "whole_step": {
                "bytes_per_particle_step": bpp,
                "radix_passes": passes,
                "achieved": bpp * value / 1e9,
                "frac": bpp * value / 1e9 / HBM_PEAK_GBS,
            },
        },
        "stage_ms_warmup_avg": {k: v[0] / max(1, v[1]) for k, v in warm.items()},
        # every stage against the same roofline (algorithmic bytes of the reference stage(s) it implements / its mean
        # HIP-event time in the warm-up)
        "per_stage_roofline": per_stage_roofline(warm, n, num_cells, not args.reference_order, real_bytes),
        "scene_build_s": t_gen,
    }
    if iisph:
        out["config"]["solver_iterations_last_step"] = s.last_iterations
    if not iisph:
        # the state of the flow the timed region ran in, at its two ends; cfl_ok: the fixed dt stayed inside the reference's own
        # CFL rule (sph.cpp:217-231: dt <= 0.4 h / (c_s + |v|max)) through the window, i.e. the timed steps integrate a stable flow
        out["developed"] = {"first_step": spin_up + args.warmup, "steps": args.steps, "ms_per_step": ms_per_step,
                            "stage_ms": {k: v[0] / max(1, v[1]) for k, v in warm.items()}, "at_start": before, "at_end": after,
                            "vmax": after["vmax"], "cfl_dt_limit": min(before["cfl_dt_limit"], after["cfl_dt_limit"]), "dt": after["dt"]}
        out["cfl_ok"] = bool(out["developed"]["cfl_dt_limit"] >= after["dt"])
        if resting:
            out["resting"] = resting
    if not args.no_cpu_baseline and not iisph:
        out["cpu_baseline"] = cpu_baseline()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
