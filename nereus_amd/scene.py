"""Deterministic synthetic dam-break scenes.

The generator is build-owned: the reference has no scene files, only `generateParticleCube`
(sph/sph.cpp:373-386) and a box sampled by the un-vendored `sph_boundary_particles` library
(main.cpp:545-546).  The SAME arrays produced here are fed to the oracle, the CPU baseline and the HIP
path, so nothing in this file is parity-relevant; it only has to be deterministic.

  fluid     lattice nx*ny*nz, spacing d = h - 0.005 (the generateParticleCube spacing), first particle at
            distance d from the walls x=y=z=0, positions computed in float64 then rounded to SReal, plus a
            uniform jitter of +-0.01*d per axis from SplitMix64(seed 0x5EED0001); w = 1; velocities 0.
  tank      x: 3 * block, y: 1.5 * block, z: 1.25 * block, rounded up to the boundary lattice.
  boundary  one layer of particles on the 5 tank faces (open top), spacing = particleRadius (0.02), on a
            global lattice so faces meet without duplicates; Akinci volumes Vb = 1 / sum_k W_poly6(x_b - x_k)
            over boundary particles within h (self included).
  grid      left to the solver: updateGrid rule of the reference (sph/sph.cpp:313-337).
"""
import numpy as np

SEED = 0x5EED0001
_GAMMA = np.uint64(0x9E3779B97F4A7C15)

# name -> lattice
CONFIGS = {
    "C1": (32, 32, 32),
    "C2": (100, 100, 100),
    "C3": (160, 160, 160),
    "C4": (256, 250, 250),
    "C5": (100, 100, 100),
    "NS": (216, 216, 216),
}


def splitmix64(seed, count):
    """First `count` outputs of SplitMix64 seeded with `seed` (vectorised)."""
    with np.errstate(over="ignore"):
        k = np.arange(1, count + 1, dtype=np.uint64)
        z = np.uint64(seed) + k * _GAMMA
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def fluid_block(nx, ny, nz, h, real=np.float32, jitter=0.01, seed=SEED, x_range=None, spacing=None):
    """Lattice block; returns pos4 (n,4) of dtype `real`.  Particle id = (ix*ny + iy)*nz + iz.

    x_range=(lo,hi) restricts generation to lattice planes lo <= ix < hi (multi-GPU ranks build only
    their share; jitter is drawn by GLOBAL particle id so the union is identical to the full scene)."""
    d = (float(real(h)) - 0.005) if spacing is None else float(spacing)
    lo, hi = (0, nx) if x_range is None else x_range
    ix = np.arange(lo, hi, dtype=np.int64)
    gx, gy, gz = np.meshgrid(ix, np.arange(ny, dtype=np.int64), np.arange(nz, dtype=np.int64), indexing="ij")
    gid = ((gx * ny + gy) * nz + gz).ravel()
    n = gid.size
    pos = np.empty((n, 4), dtype=np.float64)
    pos[:, 0] = (gx.ravel() + 1) * d
    pos[:, 1] = (gy.ravel() + 1) * d
    pos[:, 2] = (gz.ravel() + 1) * d
    pos[:, 3] = 1.0
    if jitter:
        # three draws per particle, indexed by global id: draw(3*gid + axis)
        with np.errstate(over="ignore"):
            for a in range(3):
                k = (gid.astype(np.uint64) * np.uint64(3) + np.uint64(a + 1))
                z = np.uint64(seed) + k * _GAMMA
                z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
                z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
                z = z ^ (z >> np.uint64(31))
                u = (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
                pos[:, a] += (2.0 * u - 1.0) * jitter * d
    return pos.astype(real)


def tank_extent(nx, ny, nz, h, spacing=0.02, real=np.float32):
    """Tank size (in boundary-lattice steps) for a block of nx*ny*nz."""
    d = float(real(h)) - 0.005
    tx = int(np.ceil(3.0 * nx * d / spacing))
    ty = int(np.ceil(1.5 * ny * d / spacing))
    tz = int(np.ceil(1.25 * nz * d / spacing))
    return tx, ty, tz


def boundary_box(tx, ty, tz, spacing=0.02, i_range=None):
    """Integer lattice coordinates (m,3) of the 5-face open box [0,tx]x[0,ty]x[0,tz] (top y=ty open).
    i_range=(lo,hi) keeps only lattice columns lo <= i <= hi (a rank's share of a long tank)."""
    ilo, ihi = (0, tx) if i_range is None else (max(0, i_range[0]), min(tx, i_range[1]))
    i = np.arange(ilo, ihi + 1, dtype=np.int64)
    j = np.arange(1, ty + 1, dtype=np.int64)
    k = np.arange(0, tz + 1, dtype=np.int64)
    parts = []
    a, b = np.meshgrid(i, k, indexing="ij")  # floor y=0
    parts.append(np.stack([a.ravel(), np.zeros(a.size, np.int64), b.ravel()], 1))
    a, b = np.meshgrid(j, k, indexing="ij")  # walls x=0, x=tx
    if ilo == 0:
        parts.append(np.stack([np.zeros(a.size, np.int64), a.ravel(), b.ravel()], 1))
    if ihi == tx:
        parts.append(np.stack([np.full(a.size, tx, np.int64), a.ravel(), b.ravel()], 1))
    ii = np.arange(max(1, ilo), min(tx - 1, ihi) + 1, dtype=np.int64)
    a, b = np.meshgrid(ii, j, indexing="ij")  # walls z=0, z=tz
    parts.append(np.stack([a.ravel(), b.ravel(), np.zeros(a.size, np.int64)], 1))
    parts.append(np.stack([a.ravel(), b.ravel(), np.full(a.size, tz, np.int64)], 1))
    return np.concatenate(parts, 0)


def akinci_volumes(lat, h, kpoly, spacing=0.02):
    """Vb = 1 / sum_k W_poly6(|x_b - x_k|) over boundary lattice points within h (self included)."""
    r = int(np.floor(h / spacing))
    offs = [(a, b, c) for a in range(-r, r + 1) for b in range(-r, r + 1) for c in range(-r, r + 1)
            if (a * a + b * b + c * c) * spacing * spacing < h * h]
    m = lat.shape[0]
    big = np.int64(1) << np.int64(21)
    key = (lat[:, 0] + 4) * big * big + (lat[:, 1] + 4) * big + (lat[:, 2] + 4)
    order = np.argsort(key, kind="stable")
    skey = key[order]
    acc = np.zeros(m, dtype=np.float64)
    h2 = h * h
    for (a, b, c) in offs:
        q = key + (a * big * big + b * big + c)
        pos = np.searchsorted(skey, q)
        pos[pos >= m] = m - 1
        present = skey[pos] == q
        r2 = (a * a + b * b + c * c) * spacing * spacing
        acc += present * (kpoly * (h2 - r2) ** 3)
    return 1.0 / acc


def dam_break(config="C1", h=0.0457, kpoly=None, real=np.float32, jitter=0.01, spacing=0.02, x_range=None,
              with_boundary=True):
    """Returns dict(pos, vel, bi, vbi, lattice, tank).  `kpoly` defaults to 315/(64*pi*h^9)."""
    nx, ny, nz = CONFIGS[config] if isinstance(config, str) else config
    hh = float(real(h))
    if kpoly is None:
        kpoly = 315.0 / (64.0 * np.pi * hh ** 9)
    pos = fluid_block(nx, ny, nz, h, real=real, jitter=jitter, x_range=x_range)
    vel = np.zeros_like(pos)
    out = {"pos": pos, "vel": vel, "lattice": (nx, ny, nz)}
    tx, ty, tz = tank_extent(nx, ny, nz, h, spacing, real)
    out["tank"] = (tx * spacing, ty * spacing, tz * spacing)
    if with_boundary:
        lat = boundary_box(tx, ty, tz, spacing)
        vb = akinci_volumes(lat, hh, float(kpoly), spacing)
        bi = np.empty((lat.shape[0], 4), dtype=real)
        bi[:, :3] = (lat.astype(np.float64) * spacing).astype(real)
        bi[:, 3] = 1.0
        out["bi"] = bi
        out["vbi"] = vb.astype(real)
    else:
        out["bi"] = np.zeros((0, 4), real)
        out["vbi"] = np.zeros((0,), real)
    return out
