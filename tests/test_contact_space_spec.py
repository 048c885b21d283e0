"""The contact-space solve's algorithm (tests/contact_space_spec.py, the numpy statement of csrc/nmf_dual.h) against the
oracle's optimum on the CPU: same problem, same answer, fewer and cheaper iterations."""
import numpy as np
import pytest

from contact_space_spec import solve


@pytest.fixture(scope="module")
def walking_problems(bench_blob, oracle_lib):
    """Constraint problems of a walking fly as the float64 oracle sets them up: (M, J, aref, D, qacc_smooth, warm start,
    the oracle's qacc, its iteration count, contact geoms) for 60 consecutive-ish steps."""
    from flygym_amd.controllers import TripodCPG

    fly, m = bench_blob
    o = oracle_lib.Oracle(m.to_blob(), "f64")
    o.ctrl[42:] = 1.0
    o.step(500)
    table = TripodCPG(fly.get_actuated_jointdofs_order("position"), 1e-4).targets(1, 2500)[0]
    ids = np.arange(42)
    o.step_replay(table, ids, 0, 400)
    out, k = [], 400
    for _ in range(60):
        ws = o.arr("qacc_warmstart").copy()
        o.step_replay(table, ids, k, 1); k += 1
        st = o.ints()
        if st["ncon"] == 0:
            continue
        nv, nefc = o.nv, st["nefc"]
        out.append(dict(M=o.arr("M").reshape(nv, nv).copy(), J=o.arr("J").reshape(nefc, nv).copy(), aref=o.arr("efc_aref").copy(),
                        D=o.arr("efc_D").copy(), a_s=o.arr("qacc_smooth").copy(), ws=ws, qacc=o.arr("qacc").copy(),
                        iters=st["solver_iter"], geoms=list(st["con_geom"]), active=o.arr("efc_force") > 0))
    assert len(out) >= 50
    return out


def test_contact_space_iterates_end_at_the_oracles_optimum(walking_problems):
    """float64: the optimum to 1e-9 of max |qacc| in no more eliminations than the oracle takes Newton iterations (they are
    the same iterates, row by row); float32 arithmetic of the same algorithm: 1e-4."""
    worst = {np.float64: 0.0, np.float32: 0.0}
    el, it = 0, 0
    for p in walking_problems:
        for dt in worst:
            qa, act, elim, _, stalls = solve(p["M"], p["J"], p["aref"], p["D"], p["a_s"], p["ws"], dtype=dt)
            worst[dt] = max(worst[dt], np.abs(qa - p["qacc"]).max() / np.abs(p["qacc"]).max())
            if dt is np.float64:
                assert np.array_equal(act, p["active"]) and stalls == 0
                el += elim; it += p["iters"]
    assert worst[np.float64] < 1e-9 and worst[np.float32] < 1e-4, worst
    assert el <= it + 2                                 # (the oracle's last iteration is its tolerance test)


def test_previous_active_set_is_the_better_first_guess(walking_problems):
    """Started from the previous step's final active set — matched by (geom, ordinal within the geom), new contacts by the
    start point's sign — the first elimination is usually the optimum already; never a different answer."""
    prev = {}
    el_plain = el_hist = 0
    for p in walking_problems:
        key, seen = [], {}
        for g in p["geoms"]:
            seen[g] = seen.get(g, -1) + 1
            key.append((g, seen[g]))
        qa0, act, e0, _, _ = solve(p["M"], p["J"], p["aref"], p["D"], p["a_s"], p["ws"])
        j_start = p["J"] @ p["ws"] - p["aref"]           # (fallback for contacts the history does not know: the warm start's sign pattern)
        guess = np.concatenate([prev.get(k_, j_start[4 * i:4 * i + 4] < 0) for i, k_ in enumerate(key)])
        qa1, act1, e1, _, _ = solve(p["M"], p["J"], p["aref"], p["D"], p["a_s"], p["ws"], guess=guess)
        assert np.abs(qa1 - qa0).max() < 1e-9 * np.abs(qa0).max() and np.array_equal(act, act1)
        el_plain += e0; el_hist += e1
        prev = {k_: act[4 * i:4 * i + 4] for i, k_ in enumerate(key)}
    assert el_hist < 0.75 * el_plain, (el_hist, el_plain)


def test_a_wrong_guess_costs_iterations_not_the_answer(walking_problems):
    """Any first guess — all rows active, none, or random — ends at the same optimum."""
    rng = np.random.default_rng(0)
    for p in walking_problems[::6]:
        n = len(p["D"])
        ref = solve(p["M"], p["J"], p["aref"], p["D"], p["a_s"], p["ws"])[0]
        for guess in (np.ones(n, bool), np.zeros(n, bool), rng.random(n) < 0.5):
            qa = solve(p["M"], p["J"], p["aref"], p["D"], p["a_s"], p["ws"], guess=guess)[0]
            assert np.abs(qa - ref).max() < 1e-9 * np.abs(ref).max()
        # and without the warm-start term (the hybrid kernels' flavour)
        qa = solve(p["M"], p["J"], p["aref"], p["D"], p["a_s"], p["ws"], warm=False)[0]
        assert np.abs(qa - ref).max() < 1e-9 * np.abs(ref).max()


def test_gram_blocks_and_the_column_read_give_the_oracles_A(walking_problems):
    """Round 5 replaced A's row triangle by G, the Gram matrix of the contacts' DIRECTIONS (3 per contact, one 3 x 3 block per unordered
    pair).  On the oracle's own problems: G packed in the kernel's block layout by the kernel's circulant rounds covers every word
    exactly once per pair (diagonal blocks: every word at least once, consistently), and the column read — four words of one block
    and three multiply-adds, row side or column side of the block by the contacts' order (``DualCol``) — reproduces
    A = J M^-1 J^T of the oracle's dense Jacobian, every entry of every problem (round-5 advisor: a CPU-only test of the G-to-A
    mapping and of the block addressing)."""
    from contact_space_spec import dual_col, gram_floats, pack_gram

    checked, most = 0, 0
    for p in walking_problems:
        J, M = p["J"], p["M"]
        ncon = J.shape[0] // 4
        most = max(most, ncon)
        # directions out of the pyramid rows (n + mu t1, n - mu t1, n + mu t2, n - mu t2) with the benchmark's mu = 1
        mu = np.ones(ncon)
        Jn = 0.5 * (J[0::4] + J[1::4])
        assert np.allclose(Jn, 0.5 * (J[2::4] + J[3::4]), atol=1e-12)          # the oracle's row order is the kernel's
        Jt1, Jt2 = (J[0::4] - J[1::4]) / (2 * mu[:, None]), (J[2::4] - J[3::4]) / (2 * mu[:, None])
        Jdir = np.empty((3 * ncon, J.shape[1]))
        Jdir[0::3], Jdir[1::3], Jdir[2::3] = Jn, Jt1, Jt2
        G, written = pack_gram(M, Jdir)
        assert len(G) == gram_floats(ncon) and np.isfinite(G).all() and written.min() >= 1
        A = J @ np.linalg.solve(M, J.T)
        got = np.array([[dual_col(G, mu, i, kk) for kk in range(4 * ncon)] for i in range(4 * ncon)])
        assert np.abs(got - A).max() <= 1e-12 * np.abs(A).max(), (ncon, np.abs(got - A).max())
        checked += got.size
    assert checked > 15000 and most >= 7
    assert gram_floats(16) == 1224 and gram_floats(13) == 819                      # the LDS the star / hybrid kernels give G
