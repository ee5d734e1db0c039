"""Multi-GPU slab decomposition driver: one process per GPU, torch.distributed (backend "nccl" = RCCL over
xGMI) for the neighbour exchange, libnereus_hip for everything that touches particles.

The reference is single-GPU (no NCCL/MPI/streams anywhere); this layer is new.  Protocol per
step (see csrc/nrs_kernels_slab.h for the device side):

    engine.pack(sendL, sendR)        partition current particles; fill the two message buffers
    isend/irecv with rank-1, rank+1  ONE fixed-size message per direction (migrants + 2-cell halo)
    engine.unpack(recvL, recvR)      append migrants (owned) and halo copies (read-only)
    engine.step()                    update(): density on owned + 1 cell, forces/integration on owned

Only point-to-point neighbour traffic exists; there is no collective in the data path (xGMI is a
point-to-point fabric, and the halo is O(1-5 MB) per step).  The driver is written against a small "engine"
interface so the protocol can be exercised on CPU with gloo and a checker engine (tests/test_slab_gloo.py);
the product engine is HipSlabEngine and has no CPU fallback.
"""
import os
import sys
import time

import numpy as np

from . import scene
from .params import default_params, update_grid

HALO_CELLS = 2
IISPH_HALO_CELLS = 8   # 2 * solver iterations + 4 (include/nereus_hip.h, nrs_iisph_*): the reference's minimum of 2 iterations


def iisph_halo_cells(max_iters):
    """Halo width (cells) that keeps an IISPH slab step exact for up to `max_iters` solver iterations: density 1 + displacement /
    advection 2 + 2 per Jacobi iteration + pressure force 1.  The reference's loop runs at least twice."""
    return 2 * max(2, int(max_iters)) + 4
NO_CUT_LO = -(1 << 29)
NO_CUT_HI = (1 << 29)


def cell_of(x, origin_x, cell_x, real=np.float32):
    """Unwrapped grid cell-x of positions x, evaluated as the device does (true division in SReal)."""
    x = np.asarray(x, dtype=real)
    return np.floor((x - real(origin_x)) / real(cell_x)).astype(np.int64)


def plane_cuts(nx_per_rank, world, h, origin_x, real=np.float32):
    """Cut cells for a lattice that is `world` blocks of nx_per_rank planes long: the cut between rank r-1 and r
    is the cell containing the mid-point between lattice planes r*nx-1 and r*nx.  Returns world+1 cell indices,
    open at both ends."""
    d = float(real(h)) - 0.005
    cuts = [NO_CUT_LO]
    for r in range(1, world):
        xm = (r * nx_per_rank + 0.5) * d
        cuts.append(int(cell_of([xm], origin_x, h, real)[0]) + 1)
    cuts.append(NO_CUT_HI)
    return cuts


def new_cuts(hist, old, halo, move_budget):
    """Pure function behind SlabDriver.rebalance: `hist` = owned particles per cell-x column (whole grid), `old` =
    current cuts (world+1 cell indices, the two outer ones open-ended).  Returns the new cuts."""
    world = len(old) - 1
    cum = np.concatenate([[0], np.cumsum(hist)])          # cum[c] = particles in columns < c
    total = int(cum[-1])
    new = list(old)
    for k in range(1, world):
        target = total * k // world
        want = int(np.searchsorted(cum, target, side="left"))      # smallest c with cum[c] >= target
        c = old[k]
        lo_lim = old[k - 1] + 2 * halo if k > 1 else 2 * halo
        hi_lim = old[k + 1] - 2 * halo if k < world - 1 else len(hist) - 2 * halo
        want = max(min(want, hi_lim), lo_lim)
        # walk towards `want` one column at a time while the particles changing owner fit the budget
        step = 1 if want > c else -1
        moved = 0
        while c != want:
            col = c if step > 0 else c - 1
            n_col = int(hist[col]) if 0 <= col < len(hist) else 0
            if moved + n_col > move_budget:
                break
            moved += n_col
            c += step
        new[k] = c
    for k in range(1, world):  # keep every slab at least two halos wide
        new[k] = max(new[k], (new[k - 1] if k > 1 else 0) + 2 * halo)
    return new


class HipSlabEngine:
    """Product engine: particles live in an nrs_ctx on this rank's GPU; buffers are torch CUDA tensors."""

    def __init__(self, params, capacity, msg_capacity, cell_lo, cell_hi, device_index, halo=None, iisph=False, flags=0,
                 iisph_max_iters=2):
        """halo: cells exchanged per side; None = 2 for SESPH, iisph_halo_cells(iisph_max_iters) for IISPH.  An IISPH slab run can
        do at most (halo - 4) // 2 solver iterations per step (`max_iters`; SlabDriver.iisph_step stops there and counts the step
        as truncated, like nrs_set_max_iterations on a single domain) — size the halo from the iteration budget."""
        import torch

        from . import capi

        self.torch = torch
        self.device = torch.device("cuda", device_index)
        self.msg_capacity = int(msg_capacity)
        # The solver and the exchange must run on ONE real stream: torch's default stream is the NULL stream, which
        # nrs_create takes as "make your own non-blocking stream" — and a req.wait() on torch's stream would then order
        # nothing against the solver's (the receive could still be in flight when nrs_slab_unpack reads the headers).
        self.stream = torch.cuda.Stream(device=self.device)
        assert self.stream.cuda_stream != 0
        self.iisph = bool(iisph)
        if halo is None:
            halo = iisph_halo_cells(iisph_max_iters) if iisph else HALO_CELLS
        self.max_iters = (halo - 4) // 2 if iisph else None
        self.solver = capi.Solver(params, capacity, solver=capi.IISPH if iisph else capi.SESPH, device=device_index,
                                  stream=self.stream.cuda_stream, flags=flags)
        self.cell_lo, self.cell_hi, self.halo = cell_lo, cell_hi, halo
        self.msg_bytes = self.solver.message_bytes(self.msg_capacity)
        self._configured = False

    def make_buffer(self):
        return self.torch.zeros(self.msg_bytes, dtype=self.torch.uint8, device=self.device)

    def load(self, pos, vel, bi, vbi, pres=None):
        self.solver.set_particles(pos, vel, pres)
        self.solver.set_boundaries(bi, vbi, update_grid=False)
        self.solver.slab_configure(self.cell_lo, self.cell_hi, self.halo)
        self._configured = True

    def pack(self, send_left, send_right):
        """queues the partition and returns None: the counts are read back with the received headers in unpack() (one host
        synchronisation per exchange); last_counts() has them afterwards"""
        return self.solver.slab_pack(None if send_left is None else send_left.data_ptr(),
                                     None if send_right is None else send_right.data_ptr(), self.msg_capacity, want_counts=False)

    def last_counts(self):
        return self.solver.slab_last_counts()

    def unpack(self, recv_left, recv_right):
        self.solver.slab_unpack(None if recv_left is None else recv_left.data_ptr(),
                                None if recv_right is None else recv_right.data_ptr(), self.msg_capacity)

    def step(self, k=1):
        self.solver.step(k)

    # IISPH: the solver loop is driven by SlabDriver.step (its exit test needs the average over all ranks)
    def iisph_predict(self):
        self.solver.iisph_predict()

    def iisph_iterate(self):
        return self.solver.iisph_iterate()

    def iisph_finish(self):
        self.solver.iisph_finish()

    def histogram(self, first_cell, ncells):
        """owned particles per global cell-x column (numpy int64)"""
        return self.solver.slab_histogram(first_cell, ncells).astype(np.int64)

    def set_cuts(self, cell_lo, cell_hi):
        self.cell_lo, self.cell_hi = cell_lo, cell_hi
        self.solver.slab_configure(cell_lo, cell_hi, self.halo)

    def synchronize(self):
        self.solver.synchronize()

    @property
    def n_owned(self):
        return self.solver.n_owned

    @property
    def n_local(self):
        return self.solver.n

    def owned_state(self):
        """(pos, vel) of the owned particles; valid right after pack()/unpack()."""
        pos, vel = self.solver.download()
        k = self.n_owned
        return pos[:k], vel[:k]


class SlabDriver:
    """Neighbour exchange + step loop for one rank."""

    def __init__(self, engine, rank, world, group=None, stage_through_host=False):
        import torch
        import torch.distributed as dist

        self.torch, self.dist = torch, dist
        self.engine, self.rank, self.world, self.group = engine, rank, world, group
        self.left = rank - 1 if rank > 0 else None
        self.right = rank + 1 if rank < world - 1 else None
        self.stage = stage_through_host  # gloo cannot move device tensors: bounce through host copies
        mk = engine.make_buffer
        self.send_l = mk() if self.left is not None else None
        self.recv_l = mk() if self.left is not None else None
        self.send_r = mk() if self.right is not None else None
        self.recv_r = mk() if self.right is not None else None
        self.last_counts = None

    def exchange(self):
        """pack -> one send/recv per neighbour -> unpack, all ordered on the engine's stream (when it has one)."""
        stream = getattr(self.engine, "stream", None)
        if stream is None:
            return self._exchange()
        with self.torch.cuda.stream(stream):
            return self._exchange()

    def _exchange(self):
        dist, eng = self.dist, self.engine
        counts = eng.pack(self.send_l, self.send_r)   # (None from the product engine: it does not wait here)
        ops, host = [], {}
        def wire(t):
            if not self.stage or t is None:
                return t
            h = t.cpu()
            host[id(t)] = h
            return h
        sl, sr = wire(self.send_l), wire(self.send_r)
        rl = self.recv_l if not self.stage else (None if self.recv_l is None else self.torch.empty_like(self.recv_l, device="cpu"))
        rr = self.recv_r if not self.stage else (None if self.recv_r is None else self.torch.empty_like(self.recv_r, device="cpu"))
        if self.left is not None:
            ops.append(dist.P2POp(dist.isend, sl, self.left, self.group))
            ops.append(dist.P2POp(dist.irecv, rl, self.left, self.group))
        if self.right is not None:
            ops.append(dist.P2POp(dist.isend, sr, self.right, self.group))
            ops.append(dist.P2POp(dist.irecv, rr, self.right, self.group))
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        if self.stage:
            if rl is not None:
                self.recv_l.copy_(rl)
            if rr is not None:
                self.recv_r.copy_(rr)
        eng.unpack(self.recv_l, self.recv_r)
        self.last_counts = counts if counts is not None else eng.last_counts()

    def step(self, k=1):
        for _ in range(k):
            self.exchange()
            if getattr(self.engine, "iisph", False):
                self.iisph_step()
            else:
                self.engine.step(1)

    truncated_steps = 0

    def iisph_step(self, min_iters=2):
        """IISPH::update() over the slabs: pressureSolve's loop `while ((rho_avg - 1000) > 1 || l < 2)` (sph_cuda.cu:736-741)
        with rho_avg formed from the sums of ALL ranks — one two-scalar all-reduce per iteration, the reference's
        thrust::reduce + host round trip (sph_cuda.cu:816-819) made global.  Returns the iteration count.
        The halo width bounds the iterations a slab step can do exactly (engine.max_iters = (halo - 4) // 2, the same on every
        rank): a step whose exit test still asks for more at that point is finished there and counted in `truncated_steps`
        (the multi-GPU form of nrs_set_max_iterations) instead of failing on every rank and leaving the contexts mid-step.
        min_iters: the reference's 2; tests raise it to drive more iterations than the reference's exit test ever asks for."""
        torch, dist, eng = self.torch, self.dist, self.engine
        real = np.float32 if not getattr(eng, "double", False) else np.float64
        cap = getattr(eng, "max_iters", None)
        eng.iisph_predict()
        it, rho_avg = 0, real(0)
        while (float(rho_avg) - 1000.0) > 1.0 or it < min_iters:
            if cap is not None and it >= cap:
                self.truncated_steps += 1
                break
            s, c = eng.iisph_iterate()
            t = torch.tensor([s, float(c)], dtype=torch.float64)
            dev = getattr(eng, "device", None) if dist.get_backend(self.group) == "nccl" else None
            if dev is not None:
                t = t.to(dev)
            dist.all_reduce(t, group=self.group)
            tot, cnt = (float(v) for v in t.cpu())
            rho_avg = real(real(tot) / real(cnt))   # `rho_avg = (SReal)acc; rho_avg /= N` (sph_cuda.cu:818-819)
            it += 1
        eng.iisph_finish()
        self.last_iterations = it
        return it

    def rebalance(self, grid_x, move_budget):
        """Count-balanced re-cut (a dam-break starts with all fluid in one third of the tank and then
        flows along x, so fixed cuts drift out of balance).  Call BEFORE exchange(): the new cuts must be in force when
        the next partition decides who owns what, so that the step after it evaluates every particle on exactly one
        rank.  Per-column counts of owned particles are summed over the ranks (one small all-reduce,
        every K steps, outside the per-step data path); each interior cut moves towards the column where the
        cumulative count reaches its share, but (a) by at most what `move_budget` particles allow, so the leavers fit
        the fixed message buffers of the next exchange, (b) never past its neighbours' old cuts minus two halos, so
        ownership only ever changes between adjacent ranks and no slab gets narrower than two halos.
        Returns the new (lo, hi) of this rank."""
        torch, dist, eng = self.torch, self.dist, self.engine
        hist = torch.from_numpy(eng.histogram(0, grid_x))  # owned particles may have drifted past the cuts: whole grid
        cuts = torch.zeros(self.world + 1, dtype=torch.int64)
        cuts[self.rank] = eng.cell_lo
        if self.rank == self.world - 1:
            cuts[self.world] = eng.cell_hi
        dev = getattr(eng, "device", None) if dist.get_backend(self.group) == "nccl" else None
        if dev is not None:
            hist, cuts = hist.to(dev), cuts.to(dev)
        dist.all_reduce(hist, group=self.group)
        dist.all_reduce(cuts, group=self.group)
        hist, old = hist.cpu().numpy(), [int(v) for v in cuts.cpu().numpy()]
        new = new_cuts(hist, old, eng.halo, move_budget)
        eng.set_cuts(new[self.rank], new[self.rank + 1])
        return new[self.rank], new[self.rank + 1]

    def finish(self):
        """Drop halo copies and hand over leavers one last time so that owned_state() is the true partition."""
        self.exchange()


def rank_scene(lattice_per_rank, rank, world, params, real=np.float32, spacing=0.02):
    """This rank's share of the weak-scaling dam-break: the fluid block is `world` times longer in x; every rank
    builds only the lattice planes and tank-boundary lattice columns near its slab.  Returns
    (params_with_global_grid, cuts, pos, vel, bi, vbi, global_counts)."""
    nx, ny, nz = lattice_per_rank
    h = float(params["interactionRadius"][0])
    gnx = nx * world
    tx, ty, tz = scene.tank_extent(gnx, ny, nz, h, spacing, real)
    p = params.copy()
    update_grid(p, (0.0, 0.0, 0.0), (real(tx * spacing), real(ty * spacing), real(tz * spacing)))
    ox = float(p["worldOrigin"][0][0])
    cuts = plane_cuts(nx, world, h, ox, real)
    lo, hi = cuts[rank], cuts[rank + 1]
    # fluid: my planes +-2, filtered by owning cell
    plo, phi = max(0, rank * nx - 2), min(gnx, (rank + 1) * nx + 2)
    pos = scene.fluid_block(gnx, ny, nz, h, real=real, x_range=(plo, phi))
    cx = cell_of(pos[:, 0], ox, h, real)
    pos = pos[(cx >= lo) & (cx < hi)]
    vel = np.zeros_like(pos)
    # boundary: lattice columns whose cell is within 3 cells of my slab (+3 lattice steps so the Akinci sums of the
    # kept points are complete)
    d = float(real(h)) - 0.005
    x_lo = 0.0 if rank == 0 else (plo + 1) * d - 4 * h
    x_hi = tx * spacing if rank == world - 1 else (phi) * d + 4 * h
    i_lo, i_hi = max(0, int(np.floor(x_lo / spacing)) - 3), min(tx, int(np.ceil(x_hi / spacing)) + 3)
    lat = scene.boundary_box(tx, ty, tz, spacing, i_range=(i_lo, i_hi))
    vb = scene.akinci_volumes(lat, h, float(params["kpoly"][0]), spacing)
    keep = (lat[:, 0] >= i_lo + (0 if i_lo == 0 else 3)) & (lat[:, 0] <= i_hi - (0 if i_hi == tx else 3))
    lat, vb = lat[keep], vb[keep]
    bi = np.empty((lat.shape[0], 4), dtype=real)
    bi[:, :3] = (lat.astype(np.float64) * spacing).astype(real)
    bi[:, 3] = 1.0
    return p, cuts, pos, vel, bi, vb.astype(real), dict(particles=gnx * ny * nz, tank=(tx, ty, tz))


def write_rank_scene(path, params, cell_lo, cell_hi, halo, msg_capacity, ctx_capacity, pos, vel, bi, vbi, iisph=False):
    """One rank's input for the C++ driver of the exchange (nereus_amd/host/tools/slab_rccl.cpp; fp32): header, parameter block,
    particles, boundary particles."""
    pos = np.ascontiguousarray(pos, np.float32).reshape(-1, 4)
    vel = np.ascontiguousarray(vel, np.float32).reshape(-1, 4)
    nb = 0 if bi is None else len(bi)
    p = np.array(params).reshape(1).copy()
    with open(path, "wb") as f:
        f.write(np.array([0x4C53524E, len(pos), nb], np.uint32).tobytes())
        f.write(np.array([max(cell_lo, -(1 << 30)), min(cell_hi, 1 << 30), halo], np.int32).tobytes())
        f.write(np.array([msg_capacity, ctx_capacity, 1 if iisph else 0, p.nbytes], np.uint32).tobytes())
        f.write(p.tobytes())
        f.write(pos.tobytes())
        f.write(vel.tobytes())
        if nb:
            f.write(np.ascontiguousarray(bi, np.float32).tobytes())
            f.write(np.ascontiguousarray(vbi, np.float32).tobytes())


def read_rank_result(path):
    """(pos, vel, truncated_steps, last_iterations) written by slab_rccl"""
    raw = open(path, "rb").read()
    n, trunc, iters, _ = np.frombuffer(raw[:16], np.uint32)
    a = np.frombuffer(raw[16:], np.float32).reshape(2, int(n), 4)
    return a[0].copy(), a[1].copy(), int(trunc), int(iters)


def capacities(lattice, h, n_owned, real=np.float32, headroom=1.6):
    """(message capacity, context capacity) in particles for a slab of the lattice scene: a 2-cell halo holds
    about HALO_CELLS * ny * nz * (h/d) lattice particles; 60 % head-room for compression plus migrants (a run that
    lets the column settle and the dam break — bench.py's spin-up — asks for more: the fluid at the foot of the
    8.8 m column ends up almost twice as dense as the lattice)."""
    nx, ny, nz = lattice
    d = float(real(h)) - 0.005
    halo_est = int(HALO_CELLS * ny * nz * (h / d))
    msg_cap = int(headroom * halo_est) + 8192
    return msg_cap, int(n_owned * 1.15) + 4 * msg_cap


def bench_main(args, lattice, rank, world, local_rank):
    """bench.py body for WORLD_SIZE > 1 (launched by torchrun, one rank per GPU)."""
    import torch
    import torch.distributed as dist

    from . import capi
    from .params import default_params

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # RCCL (backend "nccl") carries the device buffers directly.  NEREUS_BENCH_BACKEND=gloo (asked for explicitly) stages
    # the messages through host memory instead; a failing RCCL initialisation is an ERROR, never a silent downgrade — a
    # host-staged number must not be mistaken for an xGMI scaling result.  The line carries "backend" either way.
    backend = os.environ.get("NEREUS_BENCH_BACKEND", "nccl")
    if backend not in ("nccl", "gloo"):
        raise SystemExit("NEREUS_BENCH_BACKEND must be nccl or gloo")
    if backend == "nccl":
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    else:
        dist.init_process_group(backend="gloo")
    dev = torch.device("cuda", local_rank) if backend == "nccl" else torch.device("cpu")
    iisph = getattr(args, "solver", "sesph") == "iisph"
    max_iters = int(getattr(args, "iisph_max_iters", 2) or 2)
    halo = iisph_halo_cells(max_iters) if iisph else HALO_CELLS
    params = default_params(1 if iisph else 0)
    from bench import DEFAULT_DT, DEFAULT_SPIN_UP

    # the same window as the one-GPU line: untimed spin-up to the broken dam under a CFL-respecting time step (bench.py)
    spin_up = args.spin_up if getattr(args, "spin_up", None) is not None else (0 if iisph else DEFAULT_SPIN_UP)
    if getattr(args, "dt", None) is not None or not iisph:
        params["timestep"][0] = args.dt if getattr(args, "dt", None) is not None else DEFAULT_DT
    t_gen = time.perf_counter()
    p, cuts, pos, vel, bi, vbi, info = rank_scene(lattice, rank, world, params)
    t_gen = time.perf_counter() - t_gen
    nx, ny, nz = lattice
    msg_cap, cap = capacities(lattice, float(p["interactionRadius"][0]), len(pos), headroom=3.0 if spin_up else 1.6)
    if spin_up:
        cap += len(pos) // 4   # (count-balanced re-cuts keep the owned share near N / world; room for the drift between two re-cuts)
    if iisph:  # wider halo (8 cells instead of 2 for two iterations): halo / 2 times the halo particles per message
        msg_cap, cap = (halo // HALO_CELLS) * msg_cap, cap + (halo - HALO_CELLS) * msg_cap
    eng = HipSlabEngine(p, cap, msg_cap, cuts[rank], cuts[rank + 1], local_rank, halo=halo, iisph=iisph)
    eng.load(pos, vel, bi, vbi)
    drv = SlabDriver(eng, rank, world, stage_through_host=(backend != "nccl"))
    n_global = info["particles"]

    # NEREUS_BENCH_REBALANCE=K: count-balanced re-cut every K steps (default: every 50 steps of a run with a spin-up — the dam
    # flows along x, fixed cuts would drift out of balance and out of the message capacity; off in a run that times the resting
    # column, where the cuts have nothing to follow and the re-cut costs a histogram + an all-reduce)
    rebalance_every = int(os.environ.get("NEREUS_BENCH_REBALANCE", "50" if spin_up else "0"))
    grid_x = int(p["gridSize"][0][0])
    steps_done = 0

    def run(k):
        nonlocal steps_done
        for _ in range(k):
            if rebalance_every and world > 1 and steps_done and steps_done % rebalance_every == 0:
                drv.rebalance(grid_x, msg_cap // 4)
            drv.exchange()
            if iisph:
                drv.iisph_step()
            else:
                eng.step(1)
            steps_done += 1

    run(spin_up)
    eng.solver.set_profiling(True)
    tail = 2 if args.warmup >= 4 else 0   # (the last warm-up steps run after the stage times have been read back: see bench.py)
    run(args.warmup - tail)
    if args.warmup == 0:
        drv.exchange()  # untimed: creates the communicators (a partition without a step changes no particle)
    eng.synchronize()
    warm = eng.solver.stage_ms()
    dominant = max(warm, key=lambda k: warm[k][0]) if warm else "forces"
    dom_id = {v: k for k, v in capi.STAGE_NAMES.items()}[dominant]
    eng.solver.set_profiling([dom_id])
    if tail:
        run(tail)
        eng.synchronize()
        eng.solver.set_profiling([dom_id])

    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(args.steps)
    eng.synchronize()
    torch.cuda.synchronize()
    dist.barrier()
    dt = time.perf_counter() - t0
    dom_ms, dom_launches = eng.solver.stage_ms().get(dominant, (0.0, 0))  # HIP-event pairs of the timed steps, resolved now
    if dom_launches == 0 and dominant in warm:  # (a stage whose events belong to the partition in slab steps: keep the warm-up figure)
        dom_ms, dom_launches = warm[dominant]
    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    vm = torch.tensor([eng.solver.max_velocity()], dtype=torch.float64, device=dev)   # CFL check of the window's end state
    dist.all_reduce(vm, op=dist.ReduceOp.MAX)
    vmax = float(vm.item())
    h_, dt_, cs_ = float(p["interactionRadius"][0]), float(p["timestep"][0]), float(p["soundSpeed"][0])
    drv.finish()
    owned = torch.tensor([eng.n_owned], dtype=torch.int64, device=dev)
    dist.all_reduce(owned, op=dist.ReduceOp.SUM)
    finite = np.isfinite(eng.owned_state()[0]).all()
    ok = torch.tensor([1 if finite else 0], dtype=torch.int64, device=dev)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    if int(owned.item()) != n_global or int(ok.item()) != 1:
        raise SystemExit("slab run lost particles or went non-finite: %d of %d" % (int(owned.item()), n_global))

    from bench import FUSED_FORCES_BYTES_F32, HBM_PEAK_GBS, STAGE_BYTES_F32, measured_traffic, sesph_bytes_per_particle_step

    num_cells = int(p["numCells"][0])
    value = n_global * args.steps / dt
    bpp, passes = sesph_bytes_per_particle_step(num_cells)
    if iisph:
        from bench import iisph_stage_bytes

        ib = iisph_stage_bytes(4)
        iters = int(getattr(drv, "last_iterations", 2))
        STAGE_BYTES_F32 = dict(STAGE_BYTES_F32, **{k: v for k, v in ib.items()})
        bpp = bpp - STAGE_BYTES_F32.get("density", 0) - FUSED_FORCES_BYTES_F32 + sum(v * (iters if k == "i_solve" else 1) for k, v in ib.items())
    n_local = eng.n_local
    fused = dominant == "forces" and "integrate" not in warm
    dom_bytes = (FUSED_FORCES_BYTES_F32 if fused else STAGE_BYTES_F32.get(dominant, 0)) * n_local
    dom_avg_ms = dom_ms / max(1, dom_launches)
    achieved = (dom_bytes / (dom_avg_ms * 1e-3)) / 1e9 if dom_avg_ms > 0 else 0.0
    out = {
        "metric": "particle-steps/sec, %s dam-break" % ("IISPH" if iisph else "SESPH"),
        "value": value,
        "unit": "particle-steps/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * dt / args.steps,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "backend": "rccl" if backend == "nccl" else "gloo+host-staging",
        "config": {
            "workload": ("IISPH" if iisph else "SESPH") + " dam-break %dx%dx%d = %d particles (%d per GPU), fp32, Muller kernels, global grid %dx%dx%d, "
                        "x-slabs with a %d-cell halo exchanged per step by %s"
                        % ((nx * world, ny, nz, n_global, nx * ny * nz) + tuple(int(v) for v in p["gridSize"][0]) + (halo,) +
                           ("RCCL send/recv" if backend == "nccl" else "gloo send/recv staged through host memory (NEREUS_BENCH_BACKEND=gloo)",)),
            "particles": n_global,
            "num_cells": num_cells,
            "steps_per_s": args.steps / dt,
            "spin_up_steps": spin_up,
            "first_timed_step": spin_up + args.warmup,
            "dt": float(p["timestep"][0]),
            "rebalance_every": rebalance_every,
            "parallelism": "slab x%d" % world,
            "message_bytes_per_direction": eng.msg_bytes,
            "sort": dict(zip(("coherent_resort_steps", "fell_back_to_full_sort"), eng.solver.resort_stats())),
            **({"iisph_max_iters_per_step": eng.max_iters, "iisph_truncated_steps": drv.truncated_steps} if iisph else {}),
        },
        "roofline": {
            "bound": "hbm", "kernel": "forces+integrate+hash (one fused launch)" if fused else dominant, "achieved": achieved,
            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": measured_traffic(dominant, n_local),
            "kernel_avg_ms": dom_avg_ms,
            "kernel_launches": dom_launches, "algorithmic_bytes_per_launch": dom_bytes, "rank": 0,
            "whole_step": {"bytes_per_particle_step": bpp, "radix_passes": passes,
                           "achieved_per_gpu": bpp * value / world / 1e9, "frac_per_gpu": bpp * value / world / 1e9 / HBM_PEAK_GBS},
        },
        "scene_build_s": t_gen,
        "developed": {"first_step": spin_up + args.warmup, "steps": args.steps, "vmax": vmax, "cfl_dt_limit": 0.4 * h_ / (cs_ + vmax), "dt": dt_},
        "cfl_ok": bool(0.4 * h_ / (cs_ + vmax) >= dt_),
    }
    dist.destroy_process_group()
    return out
