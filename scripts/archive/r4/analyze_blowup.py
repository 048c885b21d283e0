"""Offline look at a state saved by gpu_blowup_probe.py (CPU only): the oracle's step, the numpy statement of the contact-space
solve in float64 / float32, the conditioning of A and of its active block."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "oracle")); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np
import oracle as orc
import flygym_amd.compose as C
from flygym_amd import make_model
from flygym_amd.utils.math import Rotation3D
from contact_space_spec import solve
kind = sys.argv[1]
fly = make_model()[0]
world = {"blocks": C.BlocksTerrainWorld, "mixed": C.MixedTerrainWorld}[kind]()
world.add_fly(fly, (0, 0, 0.8), Rotation3D("quat", (1, 0, 0, 0)))
blob = world.compile_model().to_blob()
for f in sys.argv[2:]:
    d = np.load(f)
    ids = np.arange(d["rows"].shape[1]).astype(np.int32)
    o = orc.Oracle(blob, "f64")
    for k in ("qpos", "qvel", "ctrl", "qacc_warmstart"): o.arr(k)[:] = d[k].astype(np.float64)
    ws = o.arr("qacc_warmstart").copy()
    o.step_replay(d["rows"], ids, int(d["cur"]), 1)
    st = o.ints(); nv = o.nv; nefc = st["nefc"]
    print("==", f, "world", int(d["world"]), "step", int(d["cur"]), "| oracle: ncon", st["ncon"], "geoms", st["con_geom"], "iters", st["solver_iter"], "| kernel geoms", d["geoms"][:st["ncon"] + 1].astype(int).tolist())
    M = o.arr("M").reshape(nv, nv).copy(); J = o.arr("J").reshape(nefc, nv).copy()
    aref = o.arr("efc_aref").copy(); D = o.arr("efc_D").copy(); a_s = o.arr("qacc_smooth").copy(); qacc = o.arr("qacc").copy()
    frc = o.arr("efc_force").copy()
    sc = np.abs(qacc).max()
    print(" max|qacc| oracle %.3g kernel %.3g; kernel - oracle %.3g of max" % (sc, np.abs(d["qacc_kernel"]).max(), np.abs(d["qacc_kernel"] - qacc).max() / sc))
    print(" con_dist", np.round(o.arr("con_dist"), 5).tolist())
    print(" con_pos", np.round(o.arr("con_pos").reshape(-1, 3), 4).tolist())
    print(" normals", o.arr("con_frame").reshape(-1, 9)[:, :3].tolist())
    for name, dt in (("f64", np.float64), ("f32", np.float32)):
        qa, act, el, ls, stalls = solve(M, J, aref, D, a_s, ws, dtype=dt)
        print(" spec", name, "err %.3g" % (np.abs(qa - qacc).max() / sc), "eliminations", el, "line searches", ls, "stalls", stalls)
    A = J @ np.linalg.solve(M, J.T); R = 1 / D
    act = frc > 0
    print(" active rows", act.astype(int).reshape(-1, 4).tolist())
    print(" R", np.round(R.reshape(-1, 4)[:, 0]).tolist(), "diag A", np.round(np.diag(A).reshape(-1, 4)[:, 0]).tolist())
    w = np.linalg.eigvalsh(A)
    print(" eig(A) min %.3g max %.3g; cond(A+R) %.3g; cond(active block) %.3g" % (w.min(), w.max(), np.linalg.cond(A + np.diag(R)), np.linalg.cond(A[np.ix_(act, act)] + np.diag(R[act]))))
    # float32 elimination of the full active block without pivoting, as the kernel does it: the pivots
    Aa = (A + np.diag(R)).astype(np.float32)[np.ix_(act, act)].copy()
    piv = []
    for k in range(Aa.shape[0]):
        piv.append(float(Aa[k, k]))
        l = Aa[:, k] / Aa[k, k]; l[k] = 0
        Aa -= np.outer(l, Aa[k, :]).astype(np.float32)
    print(" float32 pivots of the final active block (min %.3g):" % min(piv), np.round(piv).tolist())
