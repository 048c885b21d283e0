"""Iteration counts of the contact-space active-set Newton for different FIRST active-set guesses (numpy, float64; CPU only)."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "oracle"))
import numpy as np
import oracle as orc
from flygym_amd import make_model
from flygym_amd.controllers import TripodCPG


def solve(M, J, aref, D, a_s, a_ws, guess, max_iter=50):
    n = len(D)
    MiJt = np.linalg.solve(M, J.T); A = J @ MiJt; R = 1 / D
    e = a_ws - a_s; j0 = J @ a_s - aref; je = J @ e; eMe = e @ (M @ e)
    cost = lambda x: 0.5 * np.sum(np.where(x < 0, D * x * x, 0))
    c = 0.0 if cost(j0) < 0.5 * eMe + cost(j0 + je) else 1.0
    jar = j0 + c * je; lam = np.zeros(n)
    elim = 0; ls_count = 0
    first = True
    for it in range(max_iter):
        act = jar < 0
        if first and guess is not None:
            act = guess.copy()
        elim += 1
        idx = np.nonzero(act)[0]
        lam_t = np.zeros(n)
        if len(idx):
            S = A[np.ix_(idx, idx)] + np.diag(R[idx])
            lam_t[idx] = -np.linalg.solve(S, j0[idx])
        jar_t = j0 + A @ lam_t
        jar_t[idx] = -R[idx] * lam_t[idx]
        if np.array_equal(jar_t < 0, act):
            lam, jar, c = lam_t, jar_t, 0.0
            break
        jv = jar_t - jar; dlam = lam_t - lam; dc = -c
        Alam = jar - j0 - c * je; Adlam = jv - dc * je
        g1 = c * dc * eMe + dc * (je @ lam) + c * (je @ dlam) + dlam @ Alam
        g2 = dc * dc * eMe + 2 * dc * (je @ dlam) + dlam @ Adlam
        alpha, lo, hi = 0.0, 0.0, -1.0
        ls_count += 1
        for ls in range(50):
            x = jar + alpha * jv; m_ = x < 0
            d1 = g1 + alpha * g2 + np.sum(D[m_] * x[m_] * jv[m_]); d2 = g2 + np.sum(D[m_] * jv[m_] ** 2)
            if d2 <= 0 or d1 == 0: break
            if d1 < 0: lo = alpha
            else: hi = alpha
            nxt = alpha - d1 / d2; bis = False
            if hi >= 0 and (nxt <= lo or nxt >= hi): nxt = 0.5 * (lo + hi); bis = True
            same = (not bis) and np.array_equal((jar + alpha * jv) < 0, (jar + nxt * jv) < 0)
            ch = abs(nxt - alpha); alpha = nxt
            if same or ch <= 1e-15 * abs(nxt): break
        if alpha <= 0:
            if first and guess is not None:      # the guessed set gave no descent direction: plain Newton from here
                first = False
                continue
            break
        first = False
        lam = lam + alpha * dlam; c = c * (1 - alpha); jar = jar + alpha * jv
    qacc = a_s + c * e + MiJt @ lam
    return qacc, jar < 0, elim, ls_count


def main():
    fly, world, _ = make_model()
    m = world.compile_model()
    o = orc.Oracle(m.to_blob(), "f64")
    nv = o.nv
    o.ctrl[42:] = 1.0
    o.step(500)
    table = TripodCPG(fly.get_actuated_jointdofs_order("position"), 1e-4).targets(1, 2500)[0]
    ids = np.arange(42)
    k = 0
    o.step_replay(table, ids, 0, 300); k = 300
    tot = {}; nsteps = 0; worst = 0
    prev = {}     # (geom, slot) -> 4 bits of the last final active set
    for trial in range(1200):
        ws = o.arr("qacc_warmstart").copy()
        o.step_replay(table, ids, k, 1); k += 1
        st = o.ints()
        if st["ncon"] == 0: prev = {}; continue
        nefc = st["nefc"]
        M = o.arr("M").reshape(nv, nv).copy(); J = o.arr("J").reshape(nefc, nv).copy()
        aref = o.arr("efc_aref").copy(); D = o.arr("efc_D").copy(); a_s = o.arr("qacc_smooth").copy(); qacc = o.arr("qacc").copy()
        geoms = st["con_geom"]
        keys = []; seen = {}
        for g in geoms:
            seen[g] = seen.get(g, -1) + 1; keys.append((g, seen[g]))
        e = ws - a_s; j0 = J @ a_s - aref; je = J @ e
        jar0 = j0 + je      # sign pattern of the warm start (if chosen)
        guesses = {"newton (sign of the start point)": None}
        gn = np.zeros(nefc, bool)
        for c in range(st["ncon"]):
            if jar0[4 * c] + jar0[4 * c + 1] < 0: gn[4 * c:4 * c + 4] = True
        guesses["stick if the start point's normal residual < 0"] = gn
        gp = jar0 < 0
        for c, key in enumerate(keys):
            if key in prev: gp[4 * c:4 * c + 4] = prev[key]
        guesses["previous step's final set per (geom, slot), else sign"] = gp
        for name, g in guesses.items():
            qa, act, el, ls = solve(M, J, aref, D, a_s, ws, g)
            err = np.abs(qa - qacc).max() / np.abs(qacc).max(); worst = max(worst, err)
            t = tot.setdefault(name, [0, 0]); t[0] += el; t[1] += ls
            if name.startswith("newton"): final = act
        prev = {key: final[4 * c:4 * c + 4].copy() for c, key in enumerate(keys)}
        nsteps += 1
    print("steps", nsteps, "worst qacc err vs oracle", worst)
    for name, (el, ls) in tot.items():
        print(f"{name:60s} eliminations/step {el / nsteps:.2f}  line searches/step {ls / nsteps:.2f}")


if __name__ == "__main__":
    main()
