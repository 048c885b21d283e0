"""Vision / olfaction sensors: the numpy specification on CPU, HIP parity on MI355X."""

import numpy as np
import pytest

from flygym_amd.sensors import NUM_OMMATIDIA, RAW_IMG_HEIGHT, RAW_IMG_WIDTH, Retina, make_ommatidia_id_map


def test_id_map_is_a_721_cell_hex_lattice():
    m = make_ommatidia_id_map()
    assert m.shape == (RAW_IMG_HEIGHT, RAW_IMG_WIDTH) and m.dtype == np.int16
    counts = np.bincount(m.ravel(), minlength=NUM_OMMATIDIA + 1)
    assert m.max() == NUM_OMMATIDIA and (counts[1:] > 0).all()
    assert counts[1:].min() > 200 and counts[1:].max() < 330           # equal-area cells (~272 px)
    assert counts[0] > 0                                               # corners belong to no ommatidium
    np.testing.assert_array_equal((m > 0), (m > 0)[::-1, ::-1])        # point-symmetric footprint
    assert m[RAW_IMG_HEIGHT // 2, RAW_IMG_WIDTH // 2] == 361            # centre cell is the middle id
    r = Retina()
    assert r.num_ommatidia == 721 and 0.2 < r.pale_mask.mean() < 0.4


def test_retina_oracle_known_answers():
    import sensors_oracle as so

    r = Retina()
    white = np.full((RAW_IMG_HEIGHT, RAW_IMG_WIDTH, 3), 255, dtype=np.uint8)
    out = so.retina_resample(white, r.id_map, r.pale_mask, r.inv_norm)
    np.testing.assert_allclose(out.max(axis=1), 1.0, rtol=1e-6)
    assert ((out[:, 0] == 0) ^ (out[:, 1] == 0)).all()                # exactly one channel per ommatidium
    np.testing.assert_array_equal(out[:, 1] > 0, r.pale_mask.astype(bool))
    green = np.zeros_like(white); green[..., 1] = 200
    out = so.retina_resample(green, r.id_map, r.pale_mask, r.inv_norm)
    np.testing.assert_allclose(out[r.pale_mask == 0, 0], 200 / 255, rtol=1e-6)
    assert (out[:, 1] == 0).all()                                      # pale cells read blue = 0
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, size=(2, RAW_IMG_HEIGHT, RAW_IMG_WIDTH, 3), dtype=np.uint8)
    out = so.retina_resample(img, r.id_map, r.pale_mask, r.inv_norm)
    k = 100
    chan = 2 if r.pale_mask[k] else 1
    np.testing.assert_allclose(out[1, k, chan - 1], img[1][r.id_map == k + 1][:, chan].mean() / 255, rtol=1e-5)


@pytest.mark.gpu
def test_retina_hip_is_bit_exact():
    import torch
    import sensors_oracle as so

    r = Retina()
    rng = np.random.default_rng(7)
    imgs = rng.integers(0, 256, size=(3, 2, RAW_IMG_HEIGHT, RAW_IMG_WIDTH, 3), dtype=np.uint8)
    imgs[0, 0] = 0
    imgs[0, 1] = 255
    imgs[1, 0, ::2] = 17                                                # structured rows
    got = r.raw_image_to_hex_pxls(torch.as_tensor(imgs, device="cuda:0")).cpu().numpy()
    want = so.retina_resample(imgs, r.id_map, r.pale_mask, r.inv_norm)
    assert got.shape == (3, 2, 721, 2)
    np.testing.assert_array_equal(got, want)                           # integer sums: bit-exact
    # a tiny custom retina: 2 ommatidia over 32 pixels, ragged runs
    id_map = np.array([[0, 1, 1, 2, 2, 2, 0, 1] * 4], dtype=np.int16)
    small = Retina(id_map=id_map, pale_mask=np.array([0, 1]))
    img = rng.integers(0, 256, size=(5, 1, 32, 3), dtype=np.uint8)
    got = small.raw_image_to_hex_pxls(torch.as_tensor(img, device="cuda:0")).cpu().numpy()
    np.testing.assert_array_equal(got, so.retina_resample(img, small.id_map, small.pale_mask, small.inv_norm))
    # streaming kernel on a hostile map (2048 pixels, a multiple of 1024): runs of 1..9 pixels, so most 16-pixel chunks
    # have more than three runs (per-pixel path) and the others exercise the three-run plan, background runs included
    ids, k = [], 0
    while len(ids) < 2048:
        run = int(rng.integers(1, 10)) if k % 3 else int(rng.integers(8, 30))
        ids += [int(rng.integers(0, 41))] * run
        k += 1
    id_map = np.array(ids[:2048], dtype=np.int16).reshape(32, 64)
    id_map[0, :41] = np.arange(1, 42)[:41] % 41 + 0                      # every id 1..40 occurs at least once
    id_map[0, :40] = np.arange(1, 41)
    hostile = Retina(id_map=id_map, pale_mask=rng.integers(0, 2, 40))
    img = rng.integers(0, 256, size=(7, 32, 64, 3), dtype=np.uint8)
    got = hostile.raw_image_to_hex_pxls(torch.as_tensor(img, device="cuda:0")).cpu().numpy()
    np.testing.assert_array_equal(got, so.retina_resample(img, hostile.id_map, hostile.pale_mask, hostile.inv_norm))
    with pytest.raises(ValueError):
        r.raw_image_to_hex_pxls(torch.zeros((2, 10, 10, 3), dtype=torch.uint8, device="cuda:0"))


@pytest.mark.gpu
def test_odor_sensors_match_numpy(bench_model):
    import torch
    import sensors_oracle as so
    from flygym_amd import HIPSimulation
    from flygym_amd.sensors import ODOR_SENSOR_SITES, OdorSensors

    fly, world, _ = bench_model
    sim = HIPSimulation(world, n_worlds=5, device=0)
    sim.field("qvel")[:, :6] = torch.as_tensor(np.random.default_rng(2).normal(0, 30, (5, 6)), dtype=torch.float32, device=sim.device)
    sim.step(40)
    rng = np.random.default_rng(0)
    src = rng.uniform(-20, 20, (3, 3)); src[:, 2] = rng.uniform(0.5, 3, 3)
    peak = rng.uniform(0.1, 1.0, (3, 2))
    odor = OdorSensors(sim, fly.name, src, peak)
    got = odor.get_odor_intensities().cpu().numpy()
    names = [s.name for s in fly.get_bodysegs_order()]
    seg = [names.index(n) for n, _ in ODOR_SENSOR_SITES]
    rel = np.array([r for _, r in ODOR_SENSOR_SITES])
    want = so.odor_intensity(sim.field("seg_xpos").cpu().numpy().reshape(5, 69, 3).astype(np.float64),
                             sim.field("seg_xquat").cpu().numpy().reshape(5, 69, 4).astype(np.float64), seg, rel, src, peak)
    assert got.shape == (5, 2, 4)
    np.testing.assert_allclose(got, want, rtol=2e-5)                   # float32 vs float64, tolerance 2e-5
    assert np.abs(got[0] - got[1]).max() > 0                           # worlds moved apart
