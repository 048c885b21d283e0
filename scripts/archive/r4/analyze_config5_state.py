"""Offline look at a state saved by gpu_config5_probe.py: the oracle step, the numpy contact-space prototype in float64 and
float32 on the same matrices, the kernel variants' accelerations, the active block's condition number (CPU only)."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / 'oracle')); sys.path.insert(0, str(ROOT / 'scripts' / 'r4'))
import numpy as np
import oracle as orc
import flygym_amd.compose as C
from flygym_amd import make_model
from flygym_amd.controllers import TripodCPG
from flygym_amd.utils.math import Rotation3D
from dual_prototype import dual_solve
fly = make_model()[0]
world = C.MixedTerrainWorld(); world.add_fly(fly, (0,0,0.8), Rotation3D("quat",(1,0,0,0)))
m = world.compile_model(); blob = m.to_blob()
cpg = TripodCPG(fly.get_actuated_jointdofs_order("position"), 1e-4)
ids = np.concatenate([np.arange(42), 42 + np.arange(6)]).astype(np.int32)
for f in sys.argv[1:]:
    d = np.load(f)
    print("==", f, "errs default/nohist/primal", d["errs"], "its", d["its"])
    o = orc.Oracle(blob, "f64")
    for k in ("qpos","qvel","ctrl","qacc_warmstart"): o.arr(k)[:] = d[k].astype(np.float64)
    ws = o.arr("qacc_warmstart").copy()
    o.step_replay(d["rows"].astype(np.float64) if False else d["rows"], ids[:d["rows"].shape[1]], int(d["cur"]), 1)
    st = o.ints(); nv = o.nv; nefc = st["nefc"]
    print("ncon", st["ncon"], "geoms", st["con_geom"], "oracle iters", st["solver_iter"])
    M = o.arr("M").reshape(nv,nv).copy(); J = o.arr("J").reshape(nefc,nv).copy()
    aref = o.arr("efc_aref").copy(); D = o.arr("efc_D").copy(); a_s = o.arr("qacc_smooth").copy(); qacc = o.arr("qacc").copy()
    frc = o.arr("efc_force").copy()
    sc = np.abs(qacc).max()
    for name, dt in (("f64", np.float64), ("f32", np.float32)):
        qa, fo, it, ls = dual_solve(M, J, aref, D, a_s, ws, dt)
        print(" prototype", name, "err", np.abs(qa-qacc).max()/sc, "iters", it, "ls", ls)
    for k in ("qacc_default","qacc_nohist","qacc_primal"):
        print(" kernel", k, "err", np.abs(d[k]-qacc).max()/sc)
    A = J @ np.linalg.solve(M, J.T); R = 1/D
    act = frc > 0
    print(" active rows", act.astype(int).reshape(-1,4).tolist())
    print(" R", R.reshape(-1,4)[:,0], " diag A", np.diag(A).reshape(-1,4)[:,0])
    Aa = A[np.ix_(act,act)] + np.diag(R[act]); print(" cond(active block)", np.linalg.cond(Aa))
    # kernel's implied forces? compare qacc difference direction
    dq = d["qacc_nohist"] - qacc
    print(" nohist - oracle: max at dof", np.abs(dq).argmax(), dq[np.abs(dq).argmax()], " oracle there", qacc[np.abs(dq).argmax()])
