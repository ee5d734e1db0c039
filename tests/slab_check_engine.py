"""A checker engine for nereus_amd.slab.SlabDriver: same interface as HipSlabEngine, numpy partitioning and
the CPU oracle for the physics.  Test infrastructure only (lets the exchange protocol run under gloo on CPU)."""
import numpy as np
import torch

from nereus_amd.slab import HALO_CELLS, cell_of
from tests.oracle_lib import IISPH, SESPH, STOP_I_SOLVE, Oracle


class OracleSlabEngine:
    def __init__(self, params, msg_capacity, cell_lo, cell_hi, halo=HALO_CELLS, iisph=False, self_by_slot=False):
        self.iisph = bool(iisph)
        self.self_by_slot = bool(self_by_slot)   # NRS_FLAG_IISPH_SELF_BY_SLOT
        self.pres = np.zeros(0, np.float32)
        self._pre = None
        self._it = 0
        self.p = params.copy()
        self.cap = int(msg_capacity)
        self.lo, self.hi, self.halo = cell_lo, cell_hi, halo
        self.msg_bytes = 16 + self.cap * 32
        self.pos = np.zeros((0, 4), np.float32)
        self.vel = np.zeros((0, 4), np.float32)
        self.bi = self.vbi = None
        self._n_owned = 0
        self._ghost = (np.zeros((0, 4), np.float32), np.zeros((0, 4), np.float32))

    def make_buffer(self):
        return torch.zeros(self.msg_bytes, dtype=torch.uint8)

    def load(self, pos, vel, bi, vbi, pres=None):
        self.pos, self.vel, self.bi, self.vbi = pos.copy(), vel.copy(), bi, vbi
        self.pres = np.zeros(len(pos), np.float32) if pres is None else np.asarray(pres, np.float32).copy()
        self._n_owned = len(pos)

    def _views(self, buf):
        a = buf.numpy()
        hdr = a[:16].view(np.uint32)
        pos = a[16:16 + self.cap * 16].view(np.float32).reshape(self.cap, 4)
        vel = a[16 + self.cap * 16:16 + self.cap * 32].view(np.float32).reshape(self.cap, 4)
        return hdr, pos, vel

    def pack(self, send_left, send_right):
        ox, cs = float(self.p["worldOrigin"][0][0]), float(self.p["cellSize"][0][0])
        if self.iisph:  # the warm-start pressure travels in vel.w, as in the library (k_pressure_to_velw)
            self.vel = self.vel.copy()
            self.vel[:, 3] = self.pres
        live = self.pos[:, 3] == 1.0
        pos, vel = self.pos[live], self.vel[live]
        cx = cell_of(pos[:, 0], ox, cs)
        stay = (cx >= self.lo) & (cx < self.hi)
        mig_l, mig_r = cx < self.lo, cx >= self.hi
        halo_l = stay & (cx < self.lo + self.halo)
        halo_r = stay & (cx >= self.hi - self.halo)
        ghost = (mig_l & (cx >= self.lo - self.halo)) | (mig_r & (cx < self.hi + self.halo))
        def tag(a):
            a = a.copy()
            a[:, 3] = 2.0
            return a
        for buf, mig, hal in ((send_left, mig_l, halo_l), (send_right, mig_r, halo_r)):
            if buf is None:
                assert not mig.any(), "particles left through an end of the slab chain"
                continue
            hdr, bp, bv = self._views(buf)
            nm, nh = int(mig.sum()), int(hal.sum())
            assert nm + nh <= self.cap
            hdr[:] = (nm, nh, 0, 0)
            bp[:nm], bv[:nm] = pos[mig], vel[mig]
            bp[nm:nm + nh], bv[nm:nm + nh] = tag(pos[hal]), vel[hal]
        self._ghost = (tag(pos[ghost]), vel[ghost])
        self.pos, self.vel = pos[stay], vel[stay]
        self._n_owned = len(self.pos)
        return [int(stay.sum()), int(mig_l.sum()), int(halo_l.sum()), int(mig_r.sum()), int(halo_r.sum()), int(ghost.sum())]

    def unpack(self, recv_left, recv_right):
        mig_p, mig_v, hal_p, hal_v = [], [], [], []
        for buf in (recv_left, recv_right):
            if buf is None:
                continue
            hdr, bp, bv = self._views(buf)
            nm, nh = int(hdr[0]), int(hdr[1])
            mig_p.append(bp[:nm].copy()); mig_v.append(bv[:nm].copy())
            hal_p.append(bp[nm:nm + nh].copy()); hal_v.append(bv[nm:nm + nh].copy())
        owned_p = np.concatenate([self.pos] + mig_p)
        owned_v = np.concatenate([self.vel] + mig_v)
        self._n_owned = len(owned_p)
        self.pos = np.concatenate([owned_p, self._ghost[0]] + hal_p)
        self.vel = np.concatenate([owned_v, self._ghost[1]] + hal_v)
        if self.iisph:
            self.pres = self.vel[:, 3].copy()
            self.vel[:, 3] = 0.0

    def step(self, k=1):
        o = Oracle(self.p, solver=SESPH)
        for _ in range(k):
            o.set_particles(self.pos, self.vel)
            o.set_boundaries(self.bi, self.vbi, update_grid=False)
            o.step(1)
            self.pos, self.vel = o.get("pos"), o.get("vel")

    # ---- IISPH in phases (same interface as HipSlabEngine): the oracle has no resumable solver loop, so every call re-runs the
    #      step from the saved pre-state with a FORCED iteration count (max_iters < 0) — quadratic in the iteration count, fine
    #      for test scenes
    def _oracle(self, iters, stop):
        o = Oracle(self.p, solver=IISPH, self_by_slot=self.self_by_slot)
        pos, vel, pres = self._pre
        o.set_particles(pos, vel, pres)
        o.set_boundaries(self.bi, self.vbi, update_grid=False)
        o.step(1, stop=stop, max_iters=-iters)
        return o

    def iisph_predict(self):
        self._pre = (self.pos.copy(), self.vel.copy(), self.pres.copy())
        self._it = 0

    def iisph_iterate(self):
        self._it += 1
        o = self._oracle(self._it, STOP_I_SOLVE)
        own = o.get("sortedPos")[:, 3] == 1.0
        return float(o.get("densCorr").astype(np.float64)[own].sum()), int(own.sum())

    def iisph_finish(self):
        own = self._oracle(self._it, STOP_I_SOLVE).get("sortedPos")[:, 3] == 1.0   # halo marks, in the output (sorted) order
        o = self._oracle(self._it, 0)
        self.pos, self.vel, self.pres = o.get("pos").copy(), o.get("vel").copy(), o.get("pressure").copy()
        self.pos[~own, 3] = 2.0   # iisph_integrate sets w = 1 everywhere; a slab run keeps the halo mark (k_iisph_integrate)
        self._pre = None

    def histogram(self, first_cell, ncells):
        ox, cs = float(self.p["worldOrigin"][0][0]), float(self.p["cellSize"][0][0])
        live = self.pos[:, 3] == 1.0
        cx = cell_of(self.pos[live, 0], ox, cs) - first_cell
        cx = cx[(cx >= 0) & (cx < ncells)]
        return np.bincount(cx, minlength=ncells).astype(np.int64)

    def set_cuts(self, cell_lo, cell_hi):
        self.lo, self.hi = cell_lo, cell_hi

    @property
    def cell_lo(self):
        return self.lo

    @property
    def cell_hi(self):
        return self.hi

    def synchronize(self):
        pass

    @property
    def n_owned(self):
        return self._n_owned

    @property
    def n_local(self):
        return len(self.pos)

    def owned_state(self):
        return self.pos[:self._n_owned], self.vel[:self._n_owned]
