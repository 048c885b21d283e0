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


def test_eye_render_oracle_known_answers():
    """The renderer's specification (oracle/sensors_oracle.py::render_eye_frames) against geometry worked by hand."""
    import sensors_oracle as so

    H, W, fov, h = 512, 450, 157.0, 2.0
    sky, ground = (140, 178, 230), ((77, 77, 77), (102, 102, 102))
    down = np.eye(3)                                                    # camera axes = world axes: looks along -z (down)
    fr = so.render_eye_frames((0.5, 0.5, h), down, H, W, fov, 4.0, 0.0, sky, ground)
    assert fr.shape == (H, W, 3) and fr.dtype == np.uint8
    # straight below (0.5, 0.5): square (0, 0) -> parity 0 -> colour A; the image centre is between 4 pixels
    assert (fr[255:257, 224:226] == ground[0]).all()
    # along the middle row a pixel at rho sees the ground at x = 0.5 + h tan(rho * fov / 2): check a few columns
    for col in (260, 300, 330, 360, 400):
        rho = (col + 0.5 - W / 2) * 2 / H
        x = 0.5 + h * np.tan(rho * np.radians(fov / 2))
        want = ground[(int(np.floor(x / 4.0)) + 0) & 1]
        if abs(x / 4.0 - round(x / 4.0)) > 1e-3:                         # not on a checker edge
            assert tuple(fr[256, col]) == want, col
    # the horizon is at rho = 90 / 78.5 > 1 along the axes but inside the corners: sky only there
    rho_corner = np.hypot(W / H, 1.0)
    assert rho_corner * fov / 2 > 90
    assert tuple(fr[0, 0]) == sky and tuple(fr[0, W // 2]) != sky
    # a sphere right below: angular radius asin(r / d) -> a disc of that many pixels
    fr2 = so.render_eye_frames((0.5, 0.5, h), down, H, W, fov, 4.0, 0.0, sky, ground, [(0.5, 0.5, 1.0, 0.4)], [(10, 20, 30)])
    n_sphere = int((fr2 == (10, 20, 30)).all(axis=-1).sum())
    r_pix = np.arcsin(0.4 / 1.0) / np.radians(fov / 2) * H / 2
    assert abs(n_sphere - np.pi * r_pix ** 2) < 0.03 * np.pi * r_pix ** 2
    # the legacy eye orientations, read as extrinsic x-y-z rotations, look forward-sideways with +z up
    L = so.euler_xyz_extrinsic_to_mat((1.57, 0.0, -0.47))
    Rr = so.euler_xyz_extrinsic_to_mat((-1.57, 3.14, 0.47))
    np.testing.assert_allclose(L @ [0, 0, -1], [np.sin(0.47), np.cos(0.47), 0], atol=2e-3)
    np.testing.assert_allclose(Rr @ [0, 0, -1], [np.sin(0.47), -np.cos(0.47), 0], atol=2e-3)
    assert (L @ [0, 1, 0])[2] > 0.999 and (Rr @ [0, 1, 0])[2] > 0.999


def _world_capsules(eyes, xpos_w, xquat_w):
    import sensors_oracle as so

    caps = []
    for sg, g in zip(eyes.capsule_seg, eyes.capsule_geom.astype(np.float64)):
        Rm = so.quat_to_mat(xquat_w[sg])
        caps.append((xpos_w[sg] + Rm @ g[0:3], xpos_w[sg] + Rm @ g[3:6], g[6]))
    return caps


def test_eye_render_oracle_terrain_and_body_known_answers():
    """The relief and body parts of the renderer's specification against geometry worked by hand."""
    import sensors_oracle as so

    H, W, fov = 512, 450, 157.0
    sky, ground, wall, body = (140, 178, 230), ((77, 77, 77), (102, 102, 102)), (51, 51, 51), (120, 90, 60)
    down = np.eye(3)
    gapped = (1, (1.0, 0.3, 2.0, 0.0), 0.0)
    # straight above the middle of a 1 mm block (x = 0.5): the centre pixels see its top, i.e. the flat answer
    flat = so.render_eye_frames((0.5, 0.5, 2.0), down, H, W, fov, 4.0, 0.0, sky, ground)
    fr = so.render_eye_frames((0.5, 0.5, 2.0), down, H, W, fov, 4.0, 0.0, sky, ground, terrain=gapped, wall_rgb=wall)
    assert (fr[250:262, 219:231] == flat[250:262, 219:231]).all()
    # straight above the middle of a gap (x = 1.15, 2 mm deep, 0.3 mm wide): the centre sees the gap floor (a top surface,
    # checker colour), a ray that leaves the gap sideways before reaching the floor sees a wall
    fr = so.render_eye_frames((1.15, 0.5, 2.0), down, H, W, fov, 4.0, 0.0, sky, ground, terrain=gapped, wall_rgb=wall)
    assert tuple(fr[256, 225]) in ground
    # along the middle row (varying x): the ray with tan(angle) = 0.15 / 4 just reaches the floor's edge; beyond it -> wall
    ang_edge = np.arctan(0.15 / 4.0)
    col_edge = W / 2 + ang_edge / np.radians(fov / 2) * H / 2
    assert tuple(fr[256, int(col_edge) + 3]) == wall and tuple(fr[256, int(col_edge) - 3]) in ground
    # walls are seen only on the relief
    assert not (flat == wall).all(axis=-1).any() and (fr == wall).all(axis=-1).mean() > 0.05
    # blocks: above the corner of four squares each quadrant of the centre shows tops at two different levels -> the raised
    # squares look larger (closer): count of raised-top pixels > count of low-top pixels near the centre
    blocks = (2, (1.3, 0.35, 0.0, 0.0), 0.35)
    frb = so.render_eye_frames((1.3, 1.3, 2.0), down, H, W, fov, 100.0, 0.0, sky, ground, terrain=blocks, wall_rgb=wall)
    assert (frb == wall).all(axis=-1).any()
    # a capsule along x right below the camera: its silhouette is about (length + 2 r) x 2 r at distance 1
    cap = [((0.2, 0.5, 1.0), (0.8, 0.5, 1.0), 0.1)]
    frc = so.render_eye_frames((0.5, 0.5, 2.0), down, H, W, fov, 4.0, 0.0, sky, ground, capsules=cap, body_rgb=body)
    mask = (frc == body).all(axis=-1)
    px_per_rad = H / 2 / np.radians(fov / 2)
    want = (2 * np.arctan(0.3 / 0.9)) * (2 * np.arcsin(0.1 / 1.0)) * px_per_rad ** 2
    assert 0.75 * want < mask.sum() < 1.15 * want
    rows, cols = np.where(mask)
    assert abs(rows.mean() - 255.5) < 1.0 and abs(cols.mean() - 224.5) < 1.0 and np.ptp(cols) > 2.5 * np.ptp(rows)
    # nearest hit wins: a capsule below the ground plane is hidden
    hidden = so.render_eye_frames((0.5, 0.5, 2.0), down, H, W, fov, 4.0, 0.0, sky, ground, capsules=[((0.2, 0.5, -1.0), (0.8, 0.5, -1.0), 0.1)], body_rgb=body)
    assert (hidden == flat).all()


@pytest.mark.gpu
@pytest.mark.parametrize("world_cls", ["FlatGroundWorld", "GappedTerrainWorld", "BlocksTerrainWorld", "MixedTerrainWorld"])
def test_eye_renderer_sees_the_simulated_world(world_cls):
    """the eyes see what is simulated — the terrain relief the physics collides with and the fly's own legs,
    abdomen and wings — HIP frames vs the numpy specification, readings vs the resample of those frames."""
    import torch
    import sensors_oracle as so
    import flygym_amd.compose as C
    from flygym_amd import HIPSimulation, make_model
    from flygym_amd.utils.math import Rotation3D
    from flygym_amd.vision import EyeRenderer, Scene

    fly, world, _ = make_model()
    if world_cls != "FlatGroundWorld":
        world = getattr(C, world_cls)()
        world.add_fly(fly, (0.4, 0.1, 0.8), Rotation3D("quat", (1, 0, 0, 0)))
    n = 2
    sim = HIPSimulation(world, n_worlds=n, device=0)
    sim.field("qvel")[:, :6] = torch.as_tensor(np.random.default_rng(7).normal(0, 15, (n, 6)), dtype=torch.float32, device=sim.device)
    sim.set_leg_adhesion_states(fly.name, np.ones((n, 6), dtype=np.float32))
    sim.step(250)
    scene = Scene(spheres=[(6.0, 4.0, 1.5, 1.0)], sphere_rgb=[(0.9, 0.2, 0.1)])
    eyes = EyeRenderer(sim, fly.name, scene)
    assert len(eyes.capsule_seg) == 55
    frames, omm = eyes.render_frames(with_readings=True)
    assert torch.equal(omm, eyes.retina.raw_image_to_hex_pxls(frames)) and torch.equal(omm, eyes.render())
    names = [s.name for s in fly.get_bodysegs_order()]
    xpos = sim.field("seg_xpos").cpu().numpy().reshape(n, 69, 3).astype(np.float64)
    xquat = sim.field("seg_xquat").cpu().numpy().reshape(n, 69, 4).astype(np.float64)
    fr = frames.cpu().numpy()
    terrain = None
    if sim.model["terrain_type"][0] != 0:
        tp = sim.model["terrain_params"]
        terrain = (int(sim.model["terrain_type"][0]), tuple(float(v) for v in tp[:4]), float(tp[4]))
    body_seen = wall_seen = 0
    for w in range(n):
        caps = _world_capsules(eyes, xpos[w], xquat[w])
        for e, (seg, pos, quat) in enumerate(eyes.cameras):
            Rs = so.quat_to_mat(xquat[w, names.index(seg)])
            cam = xpos[w, names.index(seg)] + Rs @ pos
            want = so.render_eye_frames(cam, Rs @ so.quat_to_mat(quat), 512, 450, 157.0, 4.0, 0.0, scene.sky_rgb, scene.ground_rgb,
                                        scene.spheres, scene.sphere_rgb, terrain=terrain, wall_rgb=scene.wall_rgb,
                                        capsules=caps, body_rgb=scene.body_rgb)
            diff = (fr[w, e] != want).any(axis=-1).mean()
            assert diff < 5e-3, f"{world_cls} world {w} eye {e}: {diff:.2e} of the pixels differ"
            ref = so.retina_resample(want, eyes.retina.id_map, eyes.retina.pale_mask, eyes.retina.inv_norm)
            # an ommatidium averages ~300 pixels; relief multiplies the material edges (float32 rays land on either side)
            assert np.abs(omm[w, e].cpu().numpy() - ref).max() < (1e-2 if terrain is None else 3e-2)
            body_seen += int((want == np.array(scene.body_rgb, dtype=np.uint8)).all(axis=-1).sum())
            wall_seen += int((want == np.array(scene.wall_rgb, dtype=np.uint8)).all(axis=-1).sum())
    assert body_seen > 2000                                  # legs / abdomen / wings are in view
    assert (wall_seen > 500) == (terrain is not None and terrain[0] in (1, 2, 3))
    # switching both off gives back the bare scene
    bare = EyeRenderer(sim, fly.name, Scene(spheres=[(6.0, 4.0, 1.5, 1.0)], sphere_rgb=[(0.9, 0.2, 0.1)], terrain_relief=False, own_body=False))
    frb = bare.render_frames().cpu().numpy()
    assert not (frb == np.array(scene.body_rgb, dtype=np.uint8)).all(axis=-1).any()
    assert not (frb == np.array(scene.wall_rgb, dtype=np.uint8)).all(axis=-1).any()


@pytest.mark.gpu
def test_eye_renderer_matches_oracle_and_resample(bench_model):
    """HIP eye renderer: raw frames vs the numpy specification (ray-exact up to float32 rounding at material edges),
    and the fused ommatidia readings vs resampling those frames (bit-exact: integer sums)."""
    import torch
    import sensors_oracle as so
    from flygym_amd import HIPSimulation
    from flygym_amd.vision import EyeRenderer, Scene

    fly, world, _ = bench_model
    n = 3
    sim = HIPSimulation(world, n_worlds=n, device=0)
    sim.field("qvel")[:, :6] = torch.as_tensor(np.random.default_rng(5).normal(0, 20, (n, 6)), dtype=torch.float32, device=sim.device)
    sim.step(60)                                                       # three different head poses
    scene = Scene(spheres=[(6.0, 4.0, 1.5, 1.0), (5.0, -6.0, 0.8, 0.8)], sphere_rgb=[(0.05, 0.05, 0.05), (0.9, 0.2, 0.1)], own_body=False)
    eyes = EyeRenderer(sim, fly.name, scene)
    frames, omm = eyes.render_frames(with_readings=True)
    assert frames.shape == (n, 2, 512, 450, 3) and omm.shape == (n, 2, 721, 2)
    # fused readings == resample of the rendered frames, and == the readings-only call
    assert torch.equal(omm, eyes.retina.raw_image_to_hex_pxls(frames))
    assert torch.equal(omm, eyes.render())
    names = [s.name for s in fly.get_bodysegs_order()]
    xpos = sim.field("seg_xpos").cpu().numpy().reshape(n, 69, 3).astype(np.float64)
    xquat = sim.field("seg_xquat").cpu().numpy().reshape(n, 69, 4).astype(np.float64)
    fr = frames.cpu().numpy()
    for w in range(n):
        for e, (seg, pos, quat) in enumerate(eyes.cameras):
            Rs = so.quat_to_mat(xquat[w, names.index(seg)])
            cam = xpos[w, names.index(seg)] + Rs @ pos
            want = so.render_eye_frames(cam, Rs @ so.quat_to_mat(quat), 512, 450, 157.0, 4.0, 0.0, scene.sky_rgb, scene.ground_rgb,
                                        scene.spheres, scene.sphere_rgb)
            diff = (fr[w, e] != want).any(axis=-1).mean()
            assert diff < 2e-3, f"world {w} eye {e}: {diff:.2e} of the pixels differ"    # float32 rounding at edges only
            got = omm[w, e].cpu().numpy()
            ref = so.retina_resample(want, eyes.retina.id_map, eyes.retina.pale_mask, eyes.retina.inv_norm)
            assert np.abs(got - ref).max() < 5e-3
    assert (fr[0, 0] != fr[0, 1]).any() and (fr[0] != fr[1]).any()       # eyes and worlds see different things
    assert ((fr == np.array(scene.sphere_rgb[0], dtype=np.uint8)).all(axis=-1)).any()   # the dark sphere is in view somewhere
    # per-world spheres
    per_world = np.repeat(scene.spheres[None], n, axis=0).copy()
    per_world[1, 0, :3] = (6.0, 40.0, 1.5)
    eyes.set_spheres(per_world)
    fr2 = eyes.render_frames().cpu().numpy()
    assert np.array_equal(fr2[0], fr[0]) and not np.array_equal(fr2[1], fr[1])
    with pytest.raises(ValueError):
        eyes.set_spheres(np.zeros((5, 4)))


def test_sampled_pixels_specification():
    """The sampled mode's choice of rays (oracle/sensors_oracle.py::sampled_pixels): 16 pixels of every ommatidium's own
    cell, in raster order, evenly spread over the cell's pixel list — one pixel in fifteen of the lattice."""
    import sensors_oracle as so
    from flygym_amd.sensors import Retina

    r = Retina()
    px = so.sampled_pixels(r.id_map, r.num_ommatidia, 16)
    ids = r.id_map.ravel()
    assert px.shape == (721, 16) and (px >= 0).all()
    assert (ids[px] == np.arange(1, 722)[:, None]).all()                     # every ray inside its own cell
    assert (np.diff(px, axis=1) > 0).all()                                   # raster order, no pixel twice (cells have > 16 pixels)
    counts = np.bincount(ids, minlength=722)[1:]
    assert counts.min() > 16 and 13.0 < counts.sum() / px.size < 16.0        # 170 k lattice pixels (230-242 per cell), 11.5 k rays
    first = np.array([np.flatnonzero(ids == i + 1)[0] for i in range(5)])
    rank = [np.searchsorted(np.flatnonzero(ids == i + 1), px[i]) for i in range(5)]
    for i in range(5):
        np.testing.assert_array_equal(rank[i], ((2 * np.arange(16) + 1) * counts[i]) // 32)
    assert (px[:5, 0] >= first).all()
    # a flat image reads the same in both modes; a vertical edge through a cell reads within 1 / 16 + the cell's own granularity
    img = np.full((512, 450, 3), 200, dtype=np.uint8)
    full = so.retina_resample(img, r.id_map, r.pale_mask, r.inv_norm)
    samp = so.retina_sampled(img, r.id_map, r.pale_mask)
    np.testing.assert_allclose(samp, full, atol=1e-6)
    img[:, 225:] = 40
    full = so.retina_resample(img, r.id_map, r.pale_mask, r.inv_norm)
    samp = so.retina_sampled(img, r.id_map, r.pale_mask)
    assert np.abs(samp - full).max() < 0.08 and np.abs(samp - full).mean() < 2e-3


@pytest.mark.gpu
def test_eye_renderer_sampled_mode(bench_model):
    """``EyeRenderer(rays_per_ommatidium=16)``: the kernel casts the specification's 16 pixels per ommatidium and nothing else.
    Its readings equal the specification applied to the frames the pixel-exact mode renders of the same poses — bit for bit
    (integer sums of the same pixels' colours) — and approximate the pixel-exact readings as a 16-point mean does."""
    import torch
    import sensors_oracle as so
    from flygym_amd import HIPSimulation
    from flygym_amd.vision import EyeRenderer, Scene

    fly, world, _ = bench_model
    n = 4
    sim = HIPSimulation(world, n_worlds=n, device=0)
    sim.field("qvel")[:, :6] = torch.as_tensor(np.random.default_rng(7).normal(0, 20, (n, 6)), dtype=torch.float32, device=sim.device)
    sim.step(60)
    scene = Scene(spheres=[(6.0, 4.0, 1.5, 1.0), (5.0, -6.0, 0.8, 0.8)], sphere_rgb=[(0.05, 0.05, 0.05), (0.9, 0.2, 0.1)])      # own body on
    exact = EyeRenderer(sim, fly.name, scene)
    sampled = EyeRenderer(sim, fly.name, scene, rays_per_ommatidium=16)
    frames, full = exact.render_frames(with_readings=True)
    got = sampled.render()
    assert got.shape == full.shape == (n, 2, 721, 2)
    want = so.retina_sampled(frames.cpu().numpy(), exact.retina.id_map, exact.retina.pale_mask)
    assert np.array_equal(got.cpu().numpy(), want)
    err = (got - full).abs()
    assert float(err.max()) < 0.35 and float(err.mean()) < 6e-3              # a 16-point mean of cells of ~240 pixels
    assert torch.equal(got, sampled.render())                                 # deterministic
    with pytest.raises(ValueError):
        sampled.render_frames()
    with pytest.raises(ValueError):
        EyeRenderer(sim, fly.name, scene, rays_per_ommatidium=8)
