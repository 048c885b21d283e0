"""Prototype of the contact-space (dual, active-set Newton) constraint solve in numpy against the C oracle's primal Newton.
Development aid for the round-4 kernel work (CPU only).

The kernel's plan: A = J M^-1 J^T over the pyramid rows (Gram matrix of the articulated-body up-sweeps), then per iteration ONE
Gauss-Jordan elimination of [R + A | j0] with the active rows as pivots and every row taking part:
  active rows    lambda* = -x,   jar* = -R lambda*
  inactive rows  jar*    = the eliminated j0
target reached exactly when its own sign pattern equals the pivot set; otherwise an exact line search towards it.
"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "oracle"))
import numpy as np
import oracle as orc
from flygym_amd import make_model
from flygym_amd.controllers import TripodCPG


def dual_solve(M, J, aref, D, a_s, a_ws, dt=np.float32, max_iter=50, verbose=False):
    f = dt
    M = M.astype(f); J = J.astype(f); aref = aref.astype(f); D = D.astype(f); a_s = a_s.astype(f); a_ws = a_ws.astype(f)
    n = len(D)
    Minv_Jt = np.linalg.solve(M.astype(np.float64), J.T.astype(np.float64)).astype(f)      # stands in for the up-sweeps
    A = (J @ Minv_Jt).astype(f)
    R = (f(1) / D).astype(f)
    e = a_ws - a_s
    j0 = (J @ a_s - aref).astype(f)
    je = (J @ e).astype(f)
    eMe = f(e @ (M @ e))
    cost = lambda x: f(0.5) * np.sum(np.where(x < 0, D * x * x, 0)).astype(f)
    cost_ws = f(0.5) * eMe + cost(j0 + je); cost_sm = cost(j0)
    c = f(0.0) if cost_sm < cost_ws else f(1.0)
    jar = (j0 + c * je).astype(f)
    lam = np.zeros(n, f)
    iters = 0
    lsearches = 0
    for it in range(max_iter):
        act = jar < 0
        iters += 1
        # Gauss-Jordan, pivots = active rows in index order, all rows participate, column j0 rides along
        L = A.copy() + np.diag(R)
        b = j0.copy()
        for k in np.nonzero(act)[0]:
            d = L[k, k]
            l = (L[:, k] / d).astype(f); l[k] = 0
            prow = L[k, :].copy(); pb = b[k]
            L -= np.outer(l, prow).astype(f)
            b -= l * pb
        lam_t = np.where(act, -b / np.diag(L), 0).astype(f)
        jar_t = np.where(act, -R * lam_t, b).astype(f)
        if np.array_equal(jar_t < 0, act):       # the target satisfies its own active set: the optimum
            lam, jar, c = lam_t, jar_t, f(0)
            break
        jv = (jar_t - jar).astype(f)
        dlam = lam_t - lam; dc = -c
        Alam = jar - j0 - c * je
        Adlam = jv - dc * je
        g1 = c * dc * eMe + dc * (je @ lam) + c * (je @ dlam) + dlam @ Alam
        g2 = dc * dc * eMe + 2 * dc * (je @ dlam) + dlam @ Adlam
        alpha, lo, hi = f(0), f(0), f(-1)
        lsearches += 1
        for ls in range(30):
            x = jar + alpha * jv
            m_ = x < 0
            d1 = g1 + alpha * g2 + np.sum(D[m_] * x[m_] * jv[m_]); d2 = g2 + np.sum(D[m_] * jv[m_] ** 2)
            if d2 <= 0 or d1 == 0: break
            if d1 < 0: lo = alpha
            else: hi = alpha
            nxt = alpha - d1 / d2; bis = False
            if hi >= 0 and (nxt <= lo or nxt >= hi): nxt = f(0.5) * (lo + hi); bis = True
            same = (not bis) and np.array_equal((jar + alpha * jv) < 0, (jar + nxt * jv) < 0)
            ch = abs(nxt - alpha); alpha = f(nxt)
            if same or ch <= 8 * np.finfo(f).eps * abs(nxt): break
        if alpha <= 0: break
        lam = (lam + alpha * dlam).astype(f); c = f(c * (1 - alpha)); jar = (jar + alpha * jv).astype(f)
    force = np.where(jar < 0, -D * jar, 0).astype(f)
    qacc = a_s + c * e + Minv_Jt @ lam
    return qacc, force, iters, lsearches


def main():
    fly, world, _ = make_model()
    m = world.compile_model()
    o = orc.Oracle(m.to_blob(), "f64")
    nv = o.nv
    o.ctrl[42:] = 1.0
    o.step(500)
    table = TripodCPG(fly.get_actuated_jointdofs_order("position"), 1e-4).targets(1, 2500)[0]
    ids = np.arange(42)
    rng = np.random.default_rng(0)
    worst = dict(f64=0.0, f32=0.0)
    tot_it = dict(oracle=0, dual=0, ls=0); nsteps = 0
    k = 0
    o.step_replay(table, ids, 0, 300); k = 300
    for trial in range(200):
        adv = int(rng.integers(1, 9))
        o.step_replay(table, ids, k, adv); k += adv
        ws = o.arr("qacc_warmstart").copy()
        o.step_replay(table, ids, k, 1); k += 1
        st = o.ints()
        if st["ncon"] == 0: continue
        nefc = st["nefc"]
        M = o.arr("M").reshape(nv, nv).copy(); J = o.arr("J").reshape(nefc, nv).copy()
        aref = o.arr("efc_aref").copy(); D = o.arr("efc_D").copy(); a_s = o.arr("qacc_smooth").copy(); qacc = o.arr("qacc").copy()
        frc = o.arr("efc_force").copy()
        for name, dt in (("f64", np.float64), ("f32", np.float32)):
            qa, fo, it, ls = dual_solve(M, J, aref, D, a_s, ws, dt)
            err = np.abs(qa - qacc).max() / np.abs(qacc).max()
            ferr = np.abs(fo - frc).max() / max(np.abs(frc).max(), 1e-30)
            worst[name] = max(worst[name], err)
            if name == "f32":
                tot_it["dual"] += it; tot_it["ls"] += ls
                if err > 1e-3 or trial < 5:
                    print(f"trial {trial}: ncon {st['ncon']} oracle iters {st['solver_iter']} dual solves {it} (line searches {ls}) qacc err f32 {err:.2e} force err {ferr:.2e}")
        tot_it["oracle"] += st["solver_iter"]; nsteps += 1
    print("worst qacc error / max|qacc|:", worst)
    print("mean iterations: oracle", tot_it["oracle"] / nsteps, "dual solves", tot_it["dual"] / nsteps, "dual line searches", tot_it["ls"] / nsteps)


if __name__ == "__main__":
    main()
