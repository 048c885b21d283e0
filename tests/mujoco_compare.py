"""Comparison harness between the CPU oracle and a MuJoCo dump in the format of tests/golden/make_mujoco_golden.py
(records ``state??/…``).  Used by tests/test_oracle_vs_mujoco.py; TEST INFRASTRUCTURE.

``residuals`` runs the float64 oracle for one step from each recorded state under a given ``EngineSemantics`` and returns
the worst relative deviations; ``rank_semantics`` sweeps the run-time switches (pyramid_R, adhesion_contacts,
sensor_frame, max_hull_contacts) and the compile-time ones (mesh_inertia, capsule_fit, invweight0) and ranks the
combinations, so that a mismatch with real MuJoCo is answered with "flip these flags", not with a rewrite.
"""

from __future__ import annotations

import itertools

import numpy as np

RUNTIME = dict(pyramid_R=("2mu2", "plain"), adhesion_contacts=("segment_geom", "fused_body"),
               sensor_frame=("world", "contact"), max_hull_contacts=(4, 1, 2, 3))
COMPILE = dict(mesh_inertia=("exact", "convex"), capsule_fit=("inertia_box", "aabb"), invweight0=("segment", "fused_body"))


def states_of(dump) -> list[dict]:
    n = int(dump["n_states"][0])
    out = []
    for i in range(n):
        pre = f"state{i:02d}/"
        out.append({k[len(pre):]: dump[k] for k in dump.files if k.startswith(pre)} if hasattr(dump, "files") else
                   {k[len(pre):]: v for k, v in dump.items() if k.startswith(pre)})
    return out


def build_model(**semantics):
    from flygym_amd import make_model

    fly, world, _ = make_model()
    for k, v in semantics.items():
        setattr(world.semantics, k, v)
    return fly, world, world.compile_model()


def oracle_step(oracle_lib, model, rec, mode="documented"):
    o = oracle_lib.Oracle(model.to_blob(), "f64")
    o.set_solver_mode(mode)
    o.qpos[:] = rec["qpos"]; o.qvel[:] = rec["qvel"]; o.ctrl[:] = rec["ctrl"]; o.arr("qacc_warmstart")[:] = rec["qacc_warmstart"]
    o.forward()
    fwd = dict(ncon=o.ints()["ncon"], con_geom=list(o.ints()["con_geom"]), qacc=o.arr("qacc").copy(),
               qacc_smooth=o.arr("qacc_smooth").copy(), qfrc_constraint=o.arr("qfrc_constraint").copy(),
               actuator_force=o.arr("actuator_force").copy(), sensordata=o.arr("sensordata").copy(),
               con_dist=o.arr("con_dist").copy(), con_pos=o.arr("con_pos").reshape(-1, 3).copy())
    o.step(1)
    fwd["next_qpos"], fwd["next_qvel"] = o.qpos.copy(), o.qvel.copy()
    return fwd


def _rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    if a.shape != b.shape:
        return np.inf
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-12)) if a.size else 0.0


def residuals(oracle_lib, dump, **semantics) -> dict:
    """Worst deviation over the recorded states, per quantity (relative to the quantity's largest magnitude)."""
    fly, world, model = build_model(**semantics)
    seg_names = model.meta["seg_names"]
    geom_seg = [seg_names[s] for s in model["geom_seg"]]
    worst = dict(ncon_mismatch=0, contact_set_mismatch=0, qacc_smooth=0.0, qacc=0.0, actuator_force=0.0, sensor_found=0,
                 sensor_force=0.0, next_qpos=0.0)
    for rec in states_of(dump):
        got = oracle_step(oracle_lib, model, rec)
        want_segs = sorted(str(s) for s in rec["con_segment"])
        got_segs = sorted(geom_seg[g] for g in got["con_geom"])
        worst["ncon_mismatch"] += int(got["ncon"] != int(rec["ncon"][0]))
        worst["contact_set_mismatch"] += int(want_segs != got_segs)
        worst["qacc_smooth"] = max(worst["qacc_smooth"], _rel(got["qacc_smooth"], rec["qacc_smooth"]))
        worst["qacc"] = max(worst["qacc"], _rel(got["qacc"], rec["qacc"]))
        worst["actuator_force"] = max(worst["actuator_force"], _rel(got["actuator_force"], rec["actuator_force"]))
        sd_w, sd_g = np.asarray(rec["sensordata"]).reshape(-1, 16), got["sensordata"].reshape(-1, 16)
        worst["sensor_found"] += int((sd_w[:, 0] > 0).tolist() != (sd_g[:, 0] > 0).tolist())
        worst["sensor_force"] = max(worst["sensor_force"], _rel(sd_g[:, 1:4], sd_w[:, 1:4]))
        worst["next_qpos"] = max(worst["next_qpos"], float(np.abs(got["next_qpos"] - rec["next_qpos"]).max()))
    return worst


def score(res: dict) -> float:
    return (res["ncon_mismatch"] + res["contact_set_mismatch"] + res["sensor_found"]) * 1.0 + res["qacc_smooth"] + res["qacc"] \
        + res["actuator_force"] + res["sensor_force"]


def rank_semantics(oracle_lib, dump, include_compile_time=False, limit=None):
    """[(score, semantics dict, residuals)] best first, over every combination of the switches."""
    space = dict(RUNTIME)
    if include_compile_time:
        space.update(COMPILE)
    keys = list(space)
    combos = list(itertools.product(*[space[k] for k in keys]))
    if limit:
        combos = combos[:limit]
    ranked = []
    for combo in combos:
        sem = dict(zip(keys, combo))
        res = residuals(oracle_lib, dump, **sem)
        ranked.append((score(res), sem, res))
    ranked.sort(key=lambda t: t[0])
    return ranked


def oracle_as_dump(oracle_lib, n_states=4, **semantics) -> dict:
    """A dump in the MuJoCo-dump format produced by the ORACLE itself under the given semantics: lets the harness be
    exercised (and its flag search be tested) where MuJoCo is unavailable.  Not a pin."""
    fly, world, model = build_model(**semantics)
    seg_names = model.meta["seg_names"]
    o = oracle_lib.Oracle(model.to_blob(), "f64")
    o.set_solver_mode("documented")
    o.ctrl[42:] = 5.0
    o.step(450)
    out = {"n_states": np.array([n_states])}
    rng = np.random.default_rng(0)
    for i in range(n_states):
        o.ctrl[:42] = model["key_ctrl"][:42] + rng.normal(0, 0.3, 42)
        o.step(60)
        if i == n_states - 1:
            o.qpos[2] -= 0.1            # press hull geoms into the ground: multi-point manifolds
        rec = dict(qpos=o.qpos.copy(), qvel=o.qvel.copy(), ctrl=o.ctrl.copy(), qacc_warmstart=o.arr("qacc_warmstart").copy())
        got = oracle_step(oracle_lib, model, rec)
        rec.update(ncon=np.array([got["ncon"]]), con_segment=np.array([seg_names[model["geom_seg"][g]] for g in got["con_geom"]]),
                   qacc=got["qacc"], qacc_smooth=got["qacc_smooth"], actuator_force=got["actuator_force"],
                   sensordata=got["sensordata"], next_qpos=got["next_qpos"], next_qvel=got["next_qvel"])
        for k, v in rec.items():
            out[f"state{i:02d}/{k}"] = v
    return out
