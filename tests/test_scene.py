import numpy as np

from nereus_amd import scene


def test_lattice_counts_match_baseline_configs():
    assert np.prod(scene.CONFIGS["C1"]) == 32768
    assert np.prod(scene.CONFIGS["C2"]) == 1_000_000
    assert np.prod(scene.CONFIGS["C3"]) == 4_096_000
    assert np.prod(scene.CONFIGS["C4"]) == 16_000_000
    assert np.prod(scene.CONFIGS["NS"]) == 10_077_696


def test_scene_is_deterministic_and_shardable():
    a = scene.fluid_block(8, 6, 5, 0.0457)
    b = scene.fluid_block(8, 6, 5, 0.0457)
    np.testing.assert_array_equal(a, b)
    parts = [scene.fluid_block(8, 6, 5, 0.0457, x_range=(lo, hi)) for lo, hi in ((0, 3), (3, 8))]
    np.testing.assert_array_equal(np.concatenate(parts, 0), a)
    d = np.float32(0.0457) - 0.005
    assert np.all(np.abs(a[:, 0].reshape(8, -1)[0] - d) <= 0.0101 * d)
    assert np.all(a[:, 3] == 1)


def test_splitmix64_reference_values():
    # published SplitMix64 outputs for seed 0 (Vigna's reference implementation)
    z = scene.splitmix64(0, 3)
    assert [int(v) for v in z] == [0xE220A8397B1DCDAF, 0x6E789E6AA1B965F4, 0x06C45D188009454F]


def test_boundary_box_has_no_duplicates_and_open_top():
    lat = scene.boundary_box(10, 7, 5)
    assert len(np.unique(lat, axis=0)) == len(lat)
    top_interior = lat[(lat[:, 1] == 7) & (lat[:, 0] > 0) & (lat[:, 0] < 10) & (lat[:, 2] > 0) & (lat[:, 2] < 5)]
    assert len(top_interior) == 0
    vb = scene.akinci_volumes(lat, 0.0457, 315.0 / (64 * np.pi * 0.0457 ** 9))
    assert np.all(vb > 0) and np.all(np.isfinite(vb))
    # interior floor points all see the same neighbourhood
    interior = (lat[:, 1] == 0) & (lat[:, 0] > 2) & (lat[:, 0] < 8) & (lat[:, 2] == 2)
    assert np.ptp(vb[interior]) < 1e-12
