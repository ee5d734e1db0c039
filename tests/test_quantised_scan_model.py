"""CPU model of the quantised neighbour scan's SUPERSET guarantee (nereus_amd/csrc/nrs_math.h: quantize_pos, QP_MARGIN; the
threshold of Ctx::derive_kernel_params): for every pair whose exact float32 squared distance passes the density loop's cut-off,
the wrapped 10-bit integer distance must pass the integer test — at the grid origin, thousands of cells away from it (where a
float32 position is coarser than a quantum), across cell faces and for negative coordinates.  The arithmetic below restates the
device code operation by operation in numpy float32 / int64."""
import numpy as np
import pytest

QP_PER_CELL, QP_MARGIN, QP_FAR = np.float32(256.0), 2.5, 1048576.0


def quantise(pos, origin, cs):
    s = (QP_PER_CELL / cs.astype(np.float32)).astype(np.float32)                    # host: qc.s = QP_PER_CELL / (float)cellSize
    t = ((pos.astype(np.float32) - origin.astype(np.float32)).astype(np.float32) * s).astype(np.float32)   # quantize_t
    return np.floor(t).astype(np.int64) & 1023, t, s


def threshold(h, s):
    hq = float(np.max(np.float32(h) * s))                                            # derive_kernel_params
    lim = hq + QP_MARGIN
    return int(np.ceil(lim * lim)) + 1


@pytest.mark.parametrize("centre_cells", [0.0, 37.3, 1000.7, 4000.2, -250.4, 4088.5])
@pytest.mark.parametrize("cs_over_h", [1.0, 1.3, 0.53])
def test_every_exact_hit_passes_the_integer_test(centre_cells, cs_over_h):
    rng = np.random.default_rng(int(abs(centre_cells)) + int(100 * cs_over_h))
    h = np.float32(0.0457)
    cs = np.array([cs_over_h * h] * 3, np.float32)
    origin = np.array([-1.1, -1.1, -1.1], np.float32)
    centre = origin.astype(np.float64) + centre_cells * cs.astype(np.float64)
    n = 700
    pos = (centre + rng.uniform(-1.6 * h, 1.6 * h, (n, 3))).astype(np.float32)
    # some particles exactly on cell faces and a few ulps off them
    faces = origin.astype(np.float64) + np.round((pos[:60].astype(np.float64) - origin) / cs) * cs
    pos[:60] = faces.astype(np.float32)
    pos[60:90] = np.nextafter(pos[:30], np.float32(np.inf))
    k, t, s = quantise(pos, origin, cs)
    assert np.all(np.abs(t) < QP_FAR), "the scene must stay inside the range the device accepts for owners"
    assert float(h) * float(s.max()) + QP_MARGIN < 511.0
    qT = threshold(h, s)
    thr = np.float32(h) * np.float32(h)  # (the device uses the smallest T with sqrtf(T) >= h: at most an ulp above h*h)
    thr = np.nextafter(thr, np.float32(np.inf))
    missed = hits = 0
    for i in range(n):
        d = (pos[i] - pos).astype(np.float32)
        d2 = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]).astype(np.float32) + (d[:, 2] * d[:, 2]).astype(np.float32)
        exact = d2 <= thr
        dk = ((k[i] - k + 512) & 1023) - 512                                         # one subtraction with guard bits + sign extension
        sup = (dk * dk).sum(1) < qT
        hits += int(exact.sum())
        missed += int((exact & ~sup).sum())
    assert hits > n and missed == 0, (hits, missed)


def test_false_positive_rate_is_a_few_percent():
    rng = np.random.default_rng(5)
    h = np.float32(0.0457)
    cs = np.array([h] * 3, np.float32)
    origin = np.zeros(3, np.float32)
    pos = rng.uniform(0, 6 * h, (3000, 3)).astype(np.float32)
    k, t, s = quantise(pos, origin, cs)
    qT = threshold(h, s)
    exact = sup = 0
    for i in range(0, 3000, 7):
        d = (pos[i] - pos).astype(np.float64)
        e = (d * d).sum(1) < float(h) ** 2
        dk = ((k[i] - k + 512) & 1023) - 512
        near = (np.abs(d) < 2 * float(h)).all(1)   # the device only ever looks at candidates less than two cells away
        sp = ((dk * dk).sum(1) < qT) & near
        exact += int(e.sum()); sup += int(sp.sum())
    assert exact <= sup <= 1.06 * exact, (exact, sup)
