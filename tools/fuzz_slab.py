"""Randomised soak of the slab path (GPU box), all ranks as contexts of ONE process on one GPU — no process group: the neighbour
exchange is a device-to-device copy of the message buffers, everything else (nrs_slab_pack / unpack / step, in-place and
pre-classified partitions, re-cuts, the per-rank cell-table window) is the product code.  Random dam-break-like blocks, 2-4 ranks,
random cuts, random velocities that carry particles across the cuts, random re-cuts between steps; after K steps the union of the
owned particles is compared (by id) with a single-domain production run of the same scene: ids conserved, positions and velocities
within 1e-5 (the order of the particles inside a cell — hence the rounding of the sums — depends on the partition).
usage: python tools/fuzz_slab.py [seeds=50] [first=0] [sesph|iisph]      (FUZZ_SLAB_BIG=1: every rank above 32,768 particles)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nereus_amd import capi, scene, slab
from nereus_amd.params import default_params, update_grid

BIG = os.environ.get("FUZZ_SLAB_BIG") == "1"
STATS = {"migrants": 0, "merge_steps": 0, "full_sorts": 0, "particles": 0, "ranks": 0}

def rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    s = np.abs(b).max()
    return float(np.abs(a - b).max() / s) if s > 0 else float(np.abs(a - b).max())

def one(seed, iisph=False):
    rng = np.random.default_rng(seed)
    p = default_params(1 if iisph else 0).copy()
    h = float(p["interactionRadius"][0])
    world = int(rng.integers(2, 5))
    halo = slab.IISPH_HALO_CELLS if iisph else slab.HALO_CELLS
    ny, nz = int(rng.integers(10, 34)), int(rng.integers(10, 34))
    nx = int(rng.integers(world * (2 * halo + 6) + 4, world * (2 * halo + 6) + 60))
    if nx * ny * nz < 34000 and rng.random() < 0.7:   # mostly big enough for the in-place partition and the merge path
        ny, nz = max(ny, 24), max(nz, 24)
    if BIG:   # every rank above the 32,768 particles from which the partition works in place and the steps merge
        ny, nz = int(rng.integers(34, 52)), int(rng.integers(34, 52))
        nx = max(nx, int(world * 34000 / (ny * nz)) + world * (2 * halo + 6))
    sc = scene.dam_break((nx, ny, nz), h=h, kpoly=float(p["kpoly"][0]))
    pos = sc["pos"].copy(); n = len(pos)
    vel = np.zeros_like(pos)
    vel[:, 3] = np.arange(n, dtype=np.float32)                      # particle id, carried by every kernel of the path
    amp = rng.uniform(0.5, 3.0) * (0.3 if iisph else 1.0)
    vel[:, 0] = np.where(rng.random(n) < 0.5, amp, -amp).astype(np.float32)
    vel[:, 1:3] = rng.normal(0, 0.3, (n, 2)).astype(np.float32)
    p = update_grid(p, sc["bi"][:, :3].min(0), sc["bi"][:, :3].max(0))
    ox, cs, gx = float(p["worldOrigin"][0][0]), float(p["cellSize"][0][0]), int(p["gridSize"][0][0])
    cx = slab.cell_of(pos[:, 0], ox, cs)
    lo_c, hi_c = int(cx.min()), int(cx.max()) + 1
    # random interior cuts, slabs at least 2 halos + 2 cells wide
    minw = 2 * halo + 2
    for _ in range(200):
        inner = np.sort(rng.integers(lo_c + minw, hi_c - minw + 1, world - 1))
        c = [slab.NO_CUT_LO] + [int(v) for v in inner] + [slab.NO_CUT_HI]
        edges = [lo_c] + [int(v) for v in inner] + [hi_c]
        if all(edges[k + 1] - edges[k] >= minw for k in range(world)):
            break
    else:
        return "skip"
    cuts = c
    STATS["particles"] += n; STATS["ranks"] += world
    msg_cap = int(3.0 * halo * ny * nz * 1.3) + 8192
    cap = n + 4 * msg_cap
    engs, bufs = [], []
    for r in range(world):
        e = slab.HipSlabEngine(p, cap, msg_cap, cuts[r], cuts[r + 1], 0, halo=halo, iisph=iisph,
                               flags=capi.FLAG_IISPH_SELF_BY_SLOT if iisph else 0)   # (order-independent self-exclusion: DESIGN.md section 5)
        mine = (cx >= cuts[r]) & (cx < cuts[r + 1])
        e.load(pos[mine], vel[mine], sc["bi"], sc["vbi"])
        engs.append(e)
        bufs.append(dict(sl=e.make_buffer() if r > 0 else None, rl=e.make_buffer() if r > 0 else None,
                         sr=e.make_buffer() if r < world - 1 else None, rr=e.make_buffer() if r < world - 1 else None))
    steps = int(rng.integers(4, 12)) if not iisph else int(rng.integers(3, 6))
    def exchange():
        for r, e in enumerate(engs):
            with torch.cuda.stream(e.stream):
                e.pack(bufs[r]["sl"], bufs[r]["sr"])   # (does not wait: the counts are read back in unpack)
        torch.cuda.synchronize()
        for r in range(world):
            if r > 0: bufs[r]["rl"].copy_(bufs[r - 1]["sr"])
            if r < world - 1: bufs[r]["rr"].copy_(bufs[r + 1]["sl"])
        torch.cuda.synchronize()
        for r, e in enumerate(engs):
            with torch.cuda.stream(e.stream):
                e.unpack(bufs[r]["rl"], bufs[r]["rr"])
                c_ = e.last_counts()
                STATS["migrants"] += int(c_[1]) + int(c_[3])
    try:
        for it in range(steps):
            if it and rng.random() < 0.35:   # a re-cut: one interior cut moves by one or two cells (both neighbours are told before the exchange)
                k = int(rng.integers(1, world)); d = int(rng.choice([-2, -1, 1, 2]))
                new = cuts[k] + d
                left_lo = lo_c if k == 1 else cuts[k - 1]
                right_hi = hi_c if k == world - 1 else cuts[k + 1]
                if new - left_lo >= minw and right_hi - new >= minw:
                    cuts[k] = new
                    engs[k - 1].set_cuts(cuts[k - 1], cuts[k]); engs[k].set_cuts(cuts[k], cuts[k + 1])
            exchange()
            if iisph:
                for e in engs: e.iisph_predict()
                l = 0; rho_avg = np.float32(0)
                while (float(rho_avg) - 1000.0) > 1.0 or l < 2:
                    tot, cnt = 0.0, 0
                    for e in engs:
                        s_, c_ = e.iisph_iterate(); tot += s_; cnt += c_
                    rho_avg = np.float32(np.float32(tot) / np.float32(cnt)); l += 1
                    if l > 2: return "skip"      # (the 8-cell halo serves two iterations)
                for e in engs: e.iisph_finish()
            else:
                for e in engs: e.step(1)
        exchange()   # final partition: owned particles compacted, halos dropped
        torch.cuda.synchronize()
        gp = np.concatenate([e.owned_state()[0] for e in engs]); gv = np.concatenate([e.owned_state()[1] for e in engs])
    finally:
        for e in engs:
            st = e.solver.resort_stats(); STATS["merge_steps"] += int(st[0]) - int(st[1]); STATS["full_sorts"] += int(st[1])
            e.solver.close()
    s = capi.Solver(p, n, solver=capi.IISPH if iisph else capi.SESPH, flags=capi.FLAG_IISPH_SELF_BY_SLOT if iisph else 0)
    s.set_particles(pos, vel); s.set_boundaries(sc["bi"], sc["vbi"], update_grid=False)
    s.step(steps)
    rp, rv = s.download(); s.close()
    if iisph:   # iisph_integrate overwrites both w components: no id survives, pair the particles by position
        from scipy.spatial import cKDTree
        if len(gp) != n:
            return "seed %d: %d of %d particles world=%d cuts=%s lattice=%s" % (seed, len(gp), n, world, cuts[1:-1], (nx, ny, nz))
        dist_, idx = cKDTree(rp[:, :3]).query(gp[:, :3])
        if len(np.unique(idx)) != n:
            return "seed %d: particles not matched one to one (max distance %.2e) world=%d cuts=%s lattice=%s" % (seed, dist_.max(), world, cuts[1:-1], (nx, ny, nz))
        o1, o2 = np.arange(n), idx
    else:
        ids = gv[:, 3].astype(np.int64)
        if len(ids) != n or not np.array_equal(np.sort(ids), np.arange(n)):
            return "seed %d: ids not conserved (%d of %d) world=%d cuts=%s lattice=%s" % (seed, len(ids), n, world, cuts[1:-1], (nx, ny, nz))
        o1, o2 = np.argsort(ids), np.argsort(rv[:, 3].astype(np.int64))
    e_p, e_v = rel(gp[o1][:, :3], rp[o2][:, :3]), rel(gv[o1][:, :3], rv[o2][:, :3])
    tol = 1e-4 if iisph else 1e-5
    if not (e_p <= tol and e_v <= 10 * tol):
        return "seed %d: pos %.2e vel %.2e after %d steps, world=%d cuts=%s lattice=%s" % (seed, e_p, e_v, steps, world, cuts[1:-1], (nx, ny, nz))
    return None

if __name__ == "__main__":
    seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    iis = len(sys.argv) > 3 and sys.argv[3] == "iisph"
    fails = skips = 0
    for sd in range(first, first + seeds):
        r = one(sd, iis)
        if r == "skip": skips += 1
        elif r: print(r, flush=True); fails += 1
        if (sd - first) % 10 == 9: print("... %d seeds done, %d failures, %d skipped" % (sd - first + 1, fails, skips), flush=True)
    print("slab fuzz (%s): %d seeds, %d failures, %d skipped; %s" % ("iisph" if iis else "sesph", seeds, fails, skips, STATS))
    sys.exit(1 if fails else 0)
