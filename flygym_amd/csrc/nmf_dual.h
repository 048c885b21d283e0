// nmf_dual.h — the constraint solve in contact space (leg-chain kernels: a free root + identical leg chains; and the hybrid
// kernels whose data fit, see kDualS / kDualH in nmf_step.hip).
//
// Same problem and same optimum as the primal loop in physics_forward (MuJoCo's Newton solver with exact line search,
// reference src/flygym/assets/model/mujoco_globals.yaml:13-14 behind mujoco_warp.step, src/flygym/warp/simulation.py:260-263)
// — but no iteration walks the kinematic tree.  Every iterate has the form
//     qacc = qacc_smooth + c (qacc_warmstart - qacc_smooth) + M^-1 J^T lambda
// (a scalar c and one multiplier per pyramid row), so the loop lives on the rows, lane = row:
//   * once per step: A = J M^-1 J^T out of a Gram matrix.  The smooth solve's articulated-body factors (U / sqrt D, 1 / sqrt D
//     per hinge and per root axis) are kept in LDS; lane (contact, direction) pushes its unit force leaf-to-root through them
//     — u_j / sqrt(D_j) at every hinge of its leg and at the six root axes, 17 numbers — and
//     G[dir][dir'] = <root parts> + [same leg] <leg parts>:  M^-1 = L^-T D^-1 L^-1, and legs meet at the root only.
//     The vectors stay in registers (a contact's three vectors reach another contact's lanes through ds_bpermute, the
//     three pairings inside the quad through DPP); G goes to LDS as one 3x3 block per unordered pair of contacts, over buffers
//     that are dead until the next step's inertia stage.  A row of the pyramid is n +- mu t, so
//       A[(c,k)][(c',k')] = G[n,n'] + s'mu' G[n,t'] + s mu (G[t,n'] + s'mu' G[t,t'])
//     — four reads of one block and three multiply-adds when an elimination needs a column (DualCol).  Storing G instead of
//     A's row triangle is what lets 16 contacts (64 rows: the wave) fit the LDS of eight flies per CU: 9 * 16 * 17 / 2 = 1224
//     floats, where the rows' triangle took 1176 for 12 contacts and 2080 for 16 (rounds 3-4: a second kernel flavour).
//   * per iteration: ONE Gauss-Jordan elimination of [R + A | j0] (R = 1 / D, j0 = J qacc_smooth - aref) with the active
//     rows as pivots and every row taking part — see dual_eliminate.  It yields the Newton target directly: active rows
//     lambda* = -x and residual -R lambda*, inactive rows the eliminated j0.  If the target's own sign pattern is the pivot
//     set it is the optimum (KKT) and the loop ends; otherwise the exact line search towards it (rows in registers; the Gauss
//     term's derivatives are row-space dot products) and the next iteration.
//   * the first pivot set is the previous step's final active set (act_hist) where one is known.
//   * how it ends when not by the KKT test — float32 makes every cost-based test blind long before the accelerations of
//     light distal dofs are settled, and lets a tie row flip for ever: a line search without measurable descent takes the
//     active-set step (to the first row that changes sign, at most three times); the pivot set of two eliminations ago is a tie
//     and its target is taken; no step is longer than four times the way to its target; MuJoCo's improvement test and the
//     rounding floor of the cost are guards from the sixth elimination on (see the loop).
//   * once at the end: qacc from the summed row responses (one root-to-leaf pass over the kept factors; hybrid kernels: the
//     rest of the body follows the root through the smooth solve's cached factors).
// Steps with more than kDualMaxCon<TP> contacts (16: the rows are the wave's lanes; hybrid kernels 13: G in T..W), with a
// contact on the rest of the body (hybrid kernels) and tethered worlds take the primal loop; so do ALL_POSSIBLE and the
// general-tree kernels altogether.
// Code size matters here as much as instruction count: the step kernel's hot path fills the 64 KB instruction cache a pair
// of CUs shares, so what runs once per row is a loop, and the elimination — unrolled by pivot ordinal, its multipliers live in
// registers — is ordered so that only the blocks a step needs are ever fetched.
// Development switches: NMF_DUAL_DEBUG (per-iteration printf of world 0), NMF_DUAL_NOWARM (no warm-start term),
// NMF_DUAL_EXIT_FROM=<n> (first elimination at which the cost-based guards apply), NMF_NO_DUAL / NMF_NO_DUAL_HYBRID.
#pragma once

namespace nmf {

// (the factors' homes: dual_leg / dual_root / dual_aref / dual_acc in nmf_step.hip)
// G, block (c, c') with c >= c' at 9 (c (c + 1) / 2 + c'), entry [direction of c][direction of c'] (0 normal, 1 / 2 tangents):
// over Ib..W (+ dual_pad); hybrid kernels: over T..W (Ib is their one copy of the inertias)
template <class TP>
__device__ __forceinline__ float* dual_g(FlyLds<TP>& s) {
  constexpr size_t need = sizeof(float) * dual_g_floats(kDualMaxCon<TP>);
  if constexpr (kDualH<TP>) {
    static_assert(!kDualH<TP> || need <= sizeof(s.T) + sizeof(s.W), "G does not fit T..W");
    return &s.T[0][0];
  } else {
    static_assert(kDualH<TP> || need <= sizeof(s.Ib) + sizeof(s.T) + sizeof(s.W) + sizeof(s.dual_pad), "G does not fit Ib..W");
    return &s.Ib[0][0];
  }
}
// Column kk (wave-uniform) of A as row `lane` sees it: the four entries of block (contact of lane, contact of kk) that a pair of
// pyramid rows combines.  fetch() issues the reads (an elimination asks a pivot ahead), value() combines them.
struct DualCol {
  lds_cptr G;
  int cc;                 // the lane's contact
  int cc9b, blk9b;        // byte offsets: 36 cc, 36 cc (cc + 1) / 2
  int tdb, td3b;          // the lane's tangent (1: rows 0 / 1, 2: rows 2 / 3) as byte offsets 4 td, 12 td
  float smu;              // +- mu of the lane's row
  // the lane's vector as a combination of its contact's directions: alpha n + beta t (t = the lane's tangent) — a pyramid row is
  // (1, +- mu); in direction mode (dual_solve) lane k = 0 of a contact stands for n (1, 0), k = 1 / 2 for t1 / t2 (0, 1), k = 3 for nothing (0, 0)
  float alpha, beta;
  struct Raw { float nn, nt, tn, tt, smu2, a2; };
  __device__ __forceinline__ void init(const float* g, int lane, float smu_) {
    G = lds_pinned(g); cc = lane >> 2; cc9b = 36 * cc; blk9b = 36 * (cc * (cc + 1) / 2);
    const int td = 1 + ((lane >> 1) & 1);
    tdb = 4 * td; td3b = 12 * td; smu = smu_;
    alpha = 1.f; beta = smu_;
    asm("" : "+v"(cc9b), "+v"(blk9b), "+v"(tdb), "+v"(td3b));      // lane constants, not expressions to re-derive per pivot
  }
  __device__ __forceinline__ Raw fetch(int kk) const {
    // (Round 6, measured and dropped: the pivot's side of the address as three v_readlane of what lane kk holds for its own contact
    // instead of these eight scalar instructions, and the pivot search as one s_ff1: 60.0 -> 59.7 M — scalar issue is not what the
    // step is short of, vector issue is.)
    const unsigned int c2 = (unsigned int)kk >> 2, td2b = 4u + (((unsigned int)kk << 1) & 4u);      // scalar
    int c29b = (int)(36u * c2), blk29b = (int)(18u * (c2 * (c2 + 1u)));
    asm("" : "+s"(c29b), "+s"(blk29b));      // (scalar multiplies, not v_mad_u64_u32 per lane)
    // the lane's contact names the block's row side (cc >= c2) or its column side: both forms of the block's address, then
    // selected — no branch, no exec masking; the other offsets by arithmetic on the lane's two-valued multiplier
    int lo = blk9b + c29b, up = cc9b + blk29b;
    asm("" : "+v"(lo), "+v"(up));
    const bool ge = cc >= (int)c2;
    const int base = ge ? lo : up;
    const int o_nt = (ge ? 1 : 3) * (int)td2b, o_tn = ge ? td3b : tdb;
    const int a_nt = base + o_nt, a_tn = base + o_tn, a_tt = a_tn + o_nt;
    auto at = [&](int byte_off) { return *(lds_cptr)((const __attribute__((address_space(3))) char*)G + byte_off); };
    Raw r;
    r.nn = at(base); r.nt = at(a_nt); r.tn = at(a_tn); r.tt = at(a_tt);
    r.smu2 = readlane_f(beta, kk); r.a2 = readlane_f(alpha, kk);
    return r;
  }
  __device__ __forceinline__ float value(const Raw& r) const { return fmaf(beta, fmaf(r.smu2, r.tt, r.a2 * r.tn), alpha * fmaf(r.smu2, r.nt, r.a2 * r.nn)); }
};
__device__ __forceinline__ float quad_sum(float v) {
  v += NMF_DPP(v, 0xB1);
  v += NMF_DPP(v, 0x4E);
  return v;
}
constexpr int dual_root_axis(int i) { return i < 3 ? 2 - i : 8 - i; }     // elimination order of the root (aba_solve)

// Gauss-Jordan elimination of [R + A | b], left-looking, unrolled by pivot ORDINAL (the p-th active row, whatever its index):
// only the code of the pivots a step really has is ever fetched.  Lane i keeps cq[p] = (column of pivot p, row i) / sqrt(d_p),
// zero on the pivot's own row; the pivot row's entry at a later pivot column kk is that column's entry of row kk by symmetry
// of the not-yet-eliminated block — v_readlane(cq[q], kk) — so column kk of the current matrix is
//   A[.][kk] - sum_{q < p} cq[q] * cq[q](lane kk)      (two chains: the sum is a dependent sequence of multiply-adds).
// A's column kk is four reads of G (DualCol), requested a pivot ahead.  On return b holds the eliminated right-hand
// side, diag the pivot of the lane's own row (active rows).
template <int PMAX>
__device__ __forceinline__ void dual_eliminate(unsigned long long rem, const DualCol& dc, float R, int lane, float& b, float& diag) {
  float cq[PMAX];
  auto first_of = [](unsigned long long r) { return __builtin_amdgcn_readfirstlane(max(__ffsll((long long)r) - 1, 0)); };      // (0 for an empty set: a harmless fetch)
  int kk_next = first_of(rem);
  DualCol::Raw an = dc.fetch(kk_next);
  // (Round 6, measured and dropped: the two rows of a pyramid pair read the same four entries of G and differ in the sign of mu
  // only — skipping the reads and their address arithmetic when the next pivot is this one's pair mate costs a scalar branch per
  // pivot in front of the reads that are meant to be in flight early: 59.9 -> 59.0 M.)
#define NMF_DUAL_PIVOT(P)                                                                                   \
  if constexpr (P < PMAX) {                                                                                 \
    if (rem == 0ull) return;                                                                                \
    const int kk = kk_next;                                                                                 \
    rem &= rem - 1ull;                                                                                      \
    float col = dc.value(an) + (lane == kk ? R : 0.f);                                                      \
    kk_next = first_of(rem);                                                                                \
    an = dc.fetch(kk_next);                                                                                 \
    __builtin_amdgcn_sched_barrier(0);      /* the next column's reads are in flight while this pivot's chain runs */ \
    { float c1 = 0.f;                                                                                       \
      _Pragma("unroll") for (int q = 0; q + 1 < P; q += 2) {                                                \
        col = fmaf(-cq[q], readlane_f(cq[q], kk), col); c1 = fmaf(-cq[q + 1], readlane_f(cq[q + 1], kk), c1); } \
      if constexpr ((P) % 2) col = fmaf(-cq[P - 1], readlane_f(cq[P - 1], kk), col);                        \
      col += c1; }                                                                                          \
    const float d = readlane_f(col, kk);                                                                    \
    const float rs = __builtin_amdgcn_rsqf(d);                                                              \
    diag = lane == kk ? d : diag;                                                                           \
    const float cp = lane == kk ? 0.f : col * rs;                                                           \
    b = fmaf(-cp, readlane_f(b, kk) * rs, b);                                                               \
    cq[P] = cp;                                                                                             \
  }
#define NMF_DUAL_PIVOT8(P) NMF_DUAL_PIVOT(P) NMF_DUAL_PIVOT(P + 1) NMF_DUAL_PIVOT(P + 2) NMF_DUAL_PIVOT(P + 3) NMF_DUAL_PIVOT(P + 4) NMF_DUAL_PIVOT(P + 5) NMF_DUAL_PIVOT(P + 6) NMF_DUAL_PIVOT(P + 7)
  NMF_DUAL_PIVOT8(0) NMF_DUAL_PIVOT8(8) NMF_DUAL_PIVOT8(16) NMF_DUAL_PIVOT8(24) NMF_DUAL_PIVOT8(32) NMF_DUAL_PIVOT8(40) NMF_DUAL_PIVOT8(48) NMF_DUAL_PIVOT8(56)
#undef NMF_DUAL_PIVOT8
#undef NMF_DUAL_PIVOT
}

// What the primal loop needs and a contact-space solve overwrote — the inertias (star kernels: G lies on Ib; Isym holds the same
// numbers) and the twists of the unconstrained acceleration in T.  Its own function: a rejected solve is one in ten thousand, and
// inlined its registers and instructions sat in every step's way.
template <class TP>
__device__ __noinline__ void dual_restore(FlyLds<TP>& s, const GModel& m, int lane) {
  WSYNC();
  if constexpr (kDualS<TP>) {
    for (int b = lane; b < TP::NB; b += kWave) {
      const float* Q = s.Isym[b];
      float* I = s.Ib[b];
      I[0] = Q[15]; I[1] = Q[13]; I[2] = Q[5]; I[3] = Q[8]; I[4] = Q[0]; I[5] = Q[6]; I[6] = Q[11]; I[7] = Q[1]; I[8] = Q[2]; I[9] = Q[7];
    }
    WSYNC();
  }
  sweep_twists(s, s.qacc_smooth, s.T, m, lane);
}

// The CPU flavour's noslip post-pass on the rows of a contact-space solve (see dual_solve): Gauss-Seidel over the pairs of opposing
// pyramid edges with the regulariser removed; returns the rows' forces after it.  The batched flavour never runs it: on the
// leg-chain kernels it is a function of its own (dual_noslip_cold: inlined it cost the headline 1 %); the hybrid kernels, whose
// registers a call site would push into scratch (ALL_BIOLOGICAL 1 -> 10 spilled registers, 31.4 -> 30.9 M), keep it inline.
template <class TP>
__device__ __forceinline__ float dual_noslip(FlyLds<TP>& s, const GModel& m, int lane, int ncon, float frow, float j0, float R, float smu) {
  lane = opaque(lane);
  ncon = __builtin_amdgcn_readfirstlane(ncon);
  const bool on = lane < 4 * ncon;
  const float scale = 1.0f / (m.meaninertia * (float)TP::NV);
  DualCol dcol;
  dcol.init(dual_g(s), lane, smu);
    for (int sweep = 0; sweep < m.noslip_iter; ++sweep) {
      float improvement = sweep == 0 ? wave_sum(0.5f * frow * frow * R) : 0.f;       // the regulariser's share of the cost drops out
      for (int c2 = 0; c2 < ncon; ++c2) {
        for (int pp = 0; pp < 2; ++pp) {
          const int r0 = 4 * c2 + 2 * pp, r1 = r0 + 1;
          const float col0 = dcol.value(dcol.fetch(r0)), col1 = dcol.value(dcol.fetch(r1));
          const float res0 = wave_sum(on ? col0 * frow : 0.f) + readlane_f(j0, r0), res1 = wave_sum(on ? col1 * frow : 0.f) + readlane_f(j0, r1);
          const float a00 = readlane_f(col0, r0), a01 = readlane_f(col0, r1), a11 = readlane_f(col1, r1);
          const float old0 = readlane_f(frow, r0), old1 = readlane_f(frow, r1);
          const float bc0 = res0 - a00 * old0 - a01 * old1, bc1 = res1 - a01 * old0 - a11 * old1;
          const float mid = 0.5f * (old0 + old1);
          const float K1 = a00 + a11 - 2.f * a01, K0 = mid * (a00 - a11) + bc0 - bc1;
          float n0 = mid, n1 = mid;
          if (!(K1 < kMinVal)) { const float y = fminf(fmaxf(-K0 / K1, -mid), mid); n0 = mid + y; n1 = mid - y; }
          const float d0 = n0 - old0, d1 = n1 - old1;
          float change = 0.5f * (d0 * (a00 * d0 + a01 * d1) + d1 * (a01 * d0 + a11 * d1)) + d0 * res0 + d1 * res1;
          if (change > 1e-10f) { n0 = old0; n1 = old1; change = 0.f; }
          frow = lane == r0 ? n0 : lane == r1 ? n1 : frow;
          improvement -= change;
        }
      }
      if (scale * improvement < 1e-6f) break;          // noslip_tolerance (MuJoCo's default)
    }
  return frow;
}
template <class TP>
__device__ __noinline__ float dual_noslip_cold(FlyLds<TP>& s, const GModel& m, int lane, int ncon, float frow, float j0, float R, float smu) {
  return dual_noslip<TP>(s, m, lane, ncon, frow, j0, R, smu);
}

// How a solve ended (SolveReport bits, nmf_step.hip) and what the elimination it ended on violates: for every end but the
// exact one the rows in (target's sign pattern) xor (pivot set) are the target's KKT violations — the largest |residual| among
// them relative to the largest |residual| of all rows goes out as `resid` (0 = exact).
// aref of row (contact c, pyramid row k) is expected in s.vB[4 c + k] (physics_forward puts it there).  Leaves qacc and the
// contact wrenches (c_w: the Euler step's solve takes them as forces on the bodies, J^T f is never formed); returns the
// number of Newton iterations (= eliminations), the report through `report` / `resid`.
template <class TP, int NC>
__device__ __forceinline__ int dual_solve(FlyLds<TP>& s, const GModel& m, int lane, int ncon, bool walls, unsigned int& report, float& resid, float* qacc_out STAGE_ARG) {
  constexpr int NDL = TP::NDL, NLEG = TP::NLEG, SW = row_width_s<TP>();
  // (per-lane addresses of this stage are rebuilt every step: hoisted out of the persistent item loop they sat in registers
  // across all other stages and pushed 16 of the loop's other invariants into scratch — ten reloads per step)
  lane = opaque(lane);
  ncon = __builtin_amdgcn_readfirstlane(ncon);
  walls = __builtin_amdgcn_readfirstlane((int)walls) != 0;
#ifdef NMF_DUAL_NOWARM
  constexpr bool kWarm = false;
#else
  // warm-start term c e (see kDualS / kDualH).  (Round 5 measured it on the hybrid kernels too, with the leg factors in HBM and
  // vA free for e: ALL_BIOLOGICAL 2.27 -> 1.92 eliminations per step but 8 bytes over its LDS budget, seven flies per CU, 30.7 ->
  // 26.2 M; ALL_POSSIBLE 2.14 -> 1.85 eliminations, 19.1 -> 18.5 M: e's twists and e.M.e over 210 dofs cost more than they save.)
  constexpr bool kWarm = kDualS<TP>;
#endif
  const Frame fr0 = ld_frame(s, m);
  float (*const DFleg)[8] = dual_leg(s);
  float (*const DFroot)[8] = dual_root(s);
  float* const Gm = dual_g(s);
  const int n4 = 4 * ncon;
  const bool on = lane < n4;
  const int cc = on ? lane >> 2 : 0, k = lane & 3;
  const int info = s.c_info[cc];
  const int body = info_body(info);
  const V3 r = ld3(s.c_r[cc]);
  const float mu = s.c_mu[cc];
  const float D = on ? s.c_D[cc] : 0.f, R = on ? __builtin_amdgcn_rcpf(s.c_D[cc]) : 0.f;
  Frame cf = fr0;
  if constexpr (TP::kTerrain) { if (walls) cf = contact_frame(info_fid(info), fr0); }
  int leg = -1, dlast = -1;
  if (body >= TP::LB0) {
    leg = (body - TP::LB0) / TP::NBL;
    const int lb = (body - TP::LB0) % TP::NBL;
    static_for<TP::NBL>([&](auto I) { constexpr int l = decltype(I)::value; dlast += lb >= l ? TP::dofs(l) : 0; });
  }
  const int legA = leg < 0 ? 0 : leg;
  const float smu = (k & 1) ? -mu : mu;
  const V3 drow = cf.n + smu * (k < 2 ? cf.t1 : cf.t2);       // this row's direction: n +- mu t
  const SV wrow = SV{cross(r, drow), drow};                   // J_row x = wrow . (twist of the body under x)

  // the rows of this contact that were active at the end of the previous step (0: unknown), then the table is cleared for
  // this step's result
  const int hist_g = info_geom(info);
  int hist_slot = 0;
  unsigned int hist_nib = 0u;
  bool hist_any = false;
#pragma unroll
  for (int o = 1; o <= 3; ++o) hist_slot += (cc >= o && info_geom(s.c_info[cc >= o ? cc - o : 0]) == hist_g) ? 1 : 0;
  if constexpr (kDualS<TP>) hist_nib = on && hist_slot < 4 ? (s.act_hist[hist_g >> 1] >> ((hist_g & 1) * 16 + 4 * hist_slot)) & 0xfu : 0u;
  else {       // the list of the last solved step's contacts (kHistLds): the entry of the same geom and ordinal
    const unsigned int key = 0x4000u | (unsigned int)hist_g | ((unsigned int)hist_slot << 8);
#pragma unroll
    for (int i = 0; i < kHistLds<TP>; ++i) {
      const unsigned int wd = s.act_hist[i];
      if ((wd & 0x43ffu) == key) hist_nib = (wd >> 10) & 0xfu;
      if (((wd >> 16) & 0x43ffu) == key) hist_nib = (wd >> 26) & 0xfu;
    }
    if (!(on && hist_slot < 4)) hist_nib = 0u;
  }
  hist_any = __ballot(hist_nib != 0u) != 0ull;
  WSYNC();
  if (lane < kHistLds<TP>) s.act_hist[lane] = 0u;

  // ---- j0 = J qacc_smooth - aref (the smooth solve left twists(qacc_smooth) in T); warm start e = qacc_ws - qacc_smooth:
  // je = J e, eMe = e.M e
  float j0 = 0.f, je = 0.f;
  if (on) j0 = dot(wrow, ldsv(s.T[body])) - dual_aref(s)[lane];
  float eMe = 0.f;
  float c_ws;      // the scalar c: 1 = warm start, 0 = unconstrained acceleration
  float gauss, ccost;
  if constexpr (kWarm) {
    // (e for this lane's dofs stays in registers for the armature term; both turns of the wave read before the first turn stores)
    static_assert(TP::NV <= 2 * kWave, "the warm-start term keeps a lane's dofs in two registers");
    const int jb = lane + kWave;
    const bool two = jb < TP::NV;
    const float ea = lane < TP::NV ? s.qacc[lane < TP::NV ? lane : 0] - s.qacc_smooth[lane < TP::NV ? lane : 0] : 0.f;
    const float eb = two ? s.qacc[two ? jb : 0] - s.qacc_smooth[two ? jb : 0] : 0.f;
    const float arm_a = lane < TP::NV ? s.arm[lane < TP::NV ? lane : 0] : 0.f, arm_b = two ? s.arm[two ? jb : 0] : 0.f;
    if (lane < TP::NV) s.vA[lane] = ea;
    if (two) s.vA[jb] = eb;
    WSYNC();
    sweep_twists(s, s.vA, s.T, m, lane);
    if (on) je = dot(wrow, ldsv(s.T[body]));
    for (int b = lane; b < TP::NB; b += kWave) { const SV tb = ldsv(s.T[b]); eMe += dot(tb, inert_mul(s.Ib[b], tb)); }
    eMe += arm_a * ea * ea;
    eMe += arm_b * eb * eb;
    const float x1 = j0 + je;
    const float v_ws = on && x1 < 0.f ? 0.5f * D * x1 * x1 : 0.f, v_sm = on && j0 < 0.f ? 0.5f * D * j0 * j0 : 0.f;
    eMe = wave_sum(eMe);
    const float cost_ws = 0.5f * eMe + wave_sum(v_ws), cost_sm = wave_sum(v_sm);
    if (cost_sm < cost_ws) { c_ws = 0.f; gauss = 0.f; ccost = cost_sm; } else { c_ws = 1.f; gauss = 0.5f * eMe; ccost = cost_ws - 0.5f * eMe; }
  } else {       // start from the unconstrained acceleration
    c_ws = 0.f; gauss = 0.f;
    ccost = wave_sum(on && j0 < 0.f ? 0.5f * D * j0 * j0 : 0.f);
  }
  WSYNC();       // T (and, for the stars, Ib) are free from here: A goes there
  STAGE(8);

  // ---- responses of the unit forces, lane = (contact, direction n / t1 / t2)
  float ur[6], ul[NDL];
  SUB_T0();
  {
    const V3 dm = k == 0 ? cf.n : (k == 1 ? cf.t1 : cf.t2);
    const V3 tq = cross(r, dm);
    float w[6] = {tq.x, tq.y, tq.z, dm.x, dm.y, dm.z};
    const lds_cptr Sl = lds_pinned(&s.S[TP::LD0 + legA * NDL][0]);
    const lds_cptr Fl = lds_pinned(kDualGlob<TP> ? &DFroot[0][0] : &DFleg[legA * NDL][0]);      // (kDualGlob: unused)
    [[maybe_unused]] const gptr<float> Fg = G((const float*)DFleg) + legA * NDL * 8;               // kDualGlob: the factors in HBM
    static_for<NDL>([&](auto DD) {
      constexpr int d = NDL - 1 - decltype(DD)::value;
      float sj[6], f[7];
#pragma unroll
      for (int i = 0; i < 6; ++i) sj[i] = Sl[d * SW + i];
#pragma unroll
      for (int i = 0; i < 7; ++i) { if constexpr (kDualGlob<TP>) f[i] = Fg[d * 8 + i]; else f[i] = Fl[d * 8 + i]; }
      float pr = sj[0] * w[0];
#pragma unroll
      for (int i = 1; i < 6; ++i) pr = fmaf(sj[i], w[i], pr);
      const float u = d <= dlast ? pr * f[6] : 0.f;
      ul[d] = u;
#pragma unroll
      for (int i = 0; i < 6; ++i) w[i] = fmaf(-f[i], u, w[i]);
    });
    static_for<6>([&](auto II) {
      constexpr int i = decltype(II)::value, e = dual_root_axis(i);
      float f[7];
#pragma unroll
      for (int q = 0; q < 7; ++q) f[q] = DFroot[i][q];
      const float u = w[e] * f[6];
      ur[i] = u;
#pragma unroll
      for (int q = 0; q < 6; ++q) w[q] = fmaf(-f[q], u, w[q]);
    });
  }
  SUB(43);
  // ---- G = Gram matrix of the direction responses, every unordered pair of contacts once: in round t the quad of contact c
  // takes contact c - t (cyclically over the ncon contacts), whose three vectors come through ds_bpermute — lane (c, k) fetches
  // direction k — and pairs every one of them with its own: the partner's other two directions reach the lane through a
  // quad permutation (k -> k + p mod 3; lane 3 of a quad carries a copy of t2 and stores nothing).  ncon / 2 + 1 rounds.
  {
    const int kd = k < 3 ? k : 2;
    const int half = ncon >> 1;
    for (int t = 0; t <= half; ++t) {
      int c2 = cc - t;
      c2 = c2 < 0 ? c2 + ncon : c2;
      const int src = (on ? 4 * c2 + k : lane) << 2;
      auto from = [&](float v) { return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, v))); };
      float pr[6], pl[NDL];
#pragma unroll
      for (int i = 0; i < 6; ++i) pr[i] = from(ur[i]);
#pragma unroll
      for (int d = 0; d < NDL; ++d) pl[d] = from(ul[d]);
      const bool same = __builtin_amdgcn_ds_bpermute(src, leg) == leg;
      const bool ge = cc >= c2;
      int base = ge ? 9 * (cc * (cc + 1) / 2 + c2) + 3 * kd : 9 * (c2 * (c2 + 1) / 2 + cc) + kd;
      const int stride = ge ? 1 : 3;
      asm("" : "+v"(base));
      // (Round 5 tried to skip the leg parts — NDL of the 6 + NDL numbers, which couple contacts of one leg only — in the rounds
      // that cannot pair two contacts of a leg: any root-only variant of this round, as a second instantiation or under a
      // wave-uniform branch, raised the kernel's spilled registers from 12 to 50 and cost 6 %.)
      static_for<3>([&](auto PP) {
        constexpr int p = decltype(PP)::value;
        auto rot = [&](float v) {      // the fetched vector of direction (k + p) mod 3
          if constexpr (p == 0) return v; else if constexpr (p == 1) return NMF_DPP(v, 0xC9); else return NMF_DPP(v, 0xD2);
        };
        float a = rot(pr[0]) * ur[0];
#pragma unroll
        for (int i = 1; i < 6; ++i) a = fmaf(rot(pr[i]), ur[i], a);
        float bl = rot(pl[0]) * ul[0];
#pragma unroll
        for (int d = 1; d < NDL; ++d) bl = fmaf(rot(pl[d]), ul[d], bl);
        a += same ? bl : 0.f;
        const int d2 = kd + p >= 3 ? kd + p - 3 : kd + p;
        if (on && k < 3) Gm[base + stride * d2] = a;
      });
    }
  }
  WSYNC();
  SUB(44);
  STAGE(9);

  // ---- Newton iterations on the rows
  float jar = fmaf(c_ws, je, j0), lam = 0.f;
  DualCol dcol;
  dcol.init(Gm, lane, smu);
  const float scale = 1.0f / (m.meaninertia * (float)TP::NV);
  int iters = 0;
  // The first active set: the rows that were active at the end of the previous step, for the contacts that existed then (same
  // geom, same place among the geom's contacts) — an active row's residual -R lambda is small, so the start point's own sign
  // pattern mispredicts exactly the rows that matter; the previous solution's set is right about nine times in ten, and then
  // one elimination proves it.  A guess that does not even give a descent direction falls back to the start point's pattern.
  bool guessed = hist_any && !(m.solver_flags & 2);
  int stalls = 0;
  unsigned int exit_bit = kExitMaxIter;
  int npiv = 0;
  float jar_last = 0.f;                                    // the last elimination's target residuals and its violated rows
  unsigned long long viol_last = 0ull;
  unsigned long long mask_p = ~0ull, mask_pp = ~0ull;      // the pivot sets of the last two eliminations (none yet)
  unsigned long long mask = guessed ? __ballot(on && (hist_nib ? ((hist_nib >> k) & 1) != 0 : jar < 0.f)) : __ballot(on && jar < 0.f);
  for (int iter = 0; iter < m.max_iter; ++iter) {
    iters = iter + 1;
    // Gauss-Jordan on [R + A | j0]: pivots = active rows in index order, every row takes part
    // Direction pivots: a contact whose four rows are all pivots — lambda_c = E psi lies in the range of the 4 x 3 pyramid map E
    // (rows n +- mu t1, n +- mu t2), and its four equations are the three of phi = E^T lambda:
    // (G_cc + R (E^T E)^-1) phi + sum_c' G_cc' phi_c' = -(E^T E)^-1 E^T j0, E^T E = diag(4, 2 mu^2, 2 mu^2) — three pivots instead of
    // four, each column one direction of G instead of a pyramid combination.  Lanes k = 0, 1, 2 of such a contact stand for n, t1,
    // t2; the other contacts keep their rows; the matrix stays symmetric (DualCol's alpha, beta).  Same optimum.
    const unsigned int nib4 = (unsigned int)(mask >> (4 * cc)) & 0xfu;
#ifdef NMF_DUAL_DIRS
    const bool fullc = on && nib4 == 0xfu && mu > 1e-6f;
#else
    // NOT the shipped default (round 6): measured + 1.7 % (59.9 -> 60.9 M at the driver's arguments, 62.95 -> 64.0 M default, no
    // elimination beyond 47 pivots left) with the SAME one-step accuracy on identical states (scripts/onestep_error.py: median
    // 4.59e-5 against 4.61e-5 of max |qacc|, p99 2.3e-4 / 2.2e-4 over 384 states; mixed terrain 4.93e-5 / 5.03e-5) — but the
    // 150-step free rollouts of tests/test_hip_parity_r2.py leave one more chaotic clip partition behind than the row pivots do
    // (28 of 32 picks followed against 30; the bar is the primal loop's 32 minus 2), and this round's rule is that no bar moves.
    // Build with -DNMF_DUAL_DIRS to get it (the code below is the same either way: with `fullc` false every contact keeps its rows).
    const bool fullc = false;
#endif
    float b, diag = 1.f, R_e = R;
    unsigned long long mask_e = mask;
    {
      const float q0 = NMF_DPP(j0, 0x00), q1 = NMF_DPP(j0, 0x55), q2 = NMF_DPP(j0, 0xAA), q3 = NMF_DPP(j0, 0xFF);
      const float hm = 0.5f * __builtin_amdgcn_rcpf(mu);                       // 1 / (2 mu)
      const float bdir = k == 0 ? 0.25f * ((q0 + q1) + (q2 + q3)) : k == 1 ? hm * (q0 - q1) : k == 2 ? hm * (q2 - q3) : 0.f;
      b = fullc ? bdir : j0;
      R_e = fullc ? (k == 0 ? 0.25f * R : R * (2.f * hm * hm)) : R;           // R / 4, R / (2 mu^2)
      dcol.alpha = fullc ? (k == 0 ? 1.f : 0.f) : 1.f;
      dcol.beta = fullc ? ((k == 1 || k == 2) ? 1.f : 0.f) : smu;
      mask_e = __ballot(fullc ? k < 3 : (bool)((mask >> lane) & 1ull));
    }
    dual_eliminate<4 * NC>(mask_e, dcol, R_e, lane, b, diag);
    const bool act = (mask >> lane) & 1ull;
    const bool piv = (mask_e >> lane) & 1ull;
    const float x = piv ? -b * __builtin_amdgcn_rcpf(diag) : 0.f;               // phi* on direction lanes, lambda* on pivot rows
    float lam_t, jar_t;
    {
      const float x0 = NMF_DPP(x, 0x00), x1 = NMF_DPP(x, 0x55), x2 = NMF_DPP(x, 0xAA);
      const float hm = 0.5f * __builtin_amdgcn_rcpf(mu);
      const float lam_dir = fmaf((k & 1) ? -hm : hm, k < 2 ? x1 : x2, 0.25f * x0);     // E (E^T E)^-1 phi
      lam_t = fullc ? lam_dir : (act ? x : 0.f);
      jar_t = fullc ? -R * lam_dir : (act ? -R * lam_t : b);
    }
    dcol.alpha = 1.f; dcol.beta = smu;
    const unsigned long long tmask = __ballot(on && jar_t < 0.f);
    npiv = max(npiv, (int)__popcll(mask_e));
    jar_last = jar_t; viol_last = tmask ^ mask;
#ifdef NMF_DUAL_DEBUG
    {
      const float dmin = wave_min(act ? diag : 1e30f), lmax = wave_max(on ? fabsf(lam_t) : 0.f), jmax = wave_max(on ? fabsf(jar_t) : 0.f);
      const float amax = wave_max(on ? fabsf(dcol.value(dcol.fetch(0))) : 0.f);      // (column 0)
      if (blockIdx.x == 0 && lane == 0) printf("iter %d npiv %d mask %016llx tmask %016llx min pivot %g max|lam_t| %g max|jar_t| %g max diag A %g c %g\n", iter, __popcll(mask), mask, tmask, dmin, lmax, jmax, amax, c_ws);
    }
#endif
    STAGE(10);
    if (tmask == mask) {       // the target satisfies its own active set: the optimum
      lam = lam_t; jar = jar_t; c_ws = 0.f;
      exit_bit = kExitKkt;
      break;
    }
    // A tie: the pivot set of two eliminations ago again.  In exact arithmetic the cost falls with every step and no set
    // returns; in float32 a row whose residual (or force) is zero to rounding flips back and forth, and from there on the line
    // search works on differences of rounding errors (round 4's soak: a curvature of 4e-6 from two sums of 4e-5, a step of
    // 4e8, a fly leaving the scene at 100 m/s — once in 40 M steps).  Both sets' targets are the optimum but for that row:
    // this one is taken if what it violates is small against the rows' residuals.
    if (iter >= 2 && mask == mask_pp) {
      const bool off = on && (((tmask ^ mask) >> lane) & 1ull);
      const float viol = wave_max(off ? fabsf(jar_t) : 0.f), all = wave_max(on ? fabsf(jar_t) : 0.f);
      if (viol <= 1e-3f * all) { lam = lam_t; jar = jar_t; c_ws = 0.f; exit_bit = kExitTie; break; }
    }
    mask_pp = mask_p; mask_p = mask;
    // line search towards the target
    const float jv = jar_t - jar, dlam = lam_t - lam, dc = -c_ws;
    float g1, g2, s1, s2;
    {
      const float Alam = jar - j0 - c_ws * je, Adlam = fmaf(c_ws, je, jv);
      const bool neg = on && jar < 0.f;
      const float S1 = wave_sum(on ? je * lam : 0.f), S2 = wave_sum(on ? je * dlam : 0.f);
      const float S3 = wave_sum(on ? dlam * Alam : 0.f), S4 = wave_sum(on ? dlam * Adlam : 0.f);
      s1 = wave_sum(neg ? D * jar * jv : 0.f); s2 = wave_sum(neg ? D * jv * jv : 0.f);
      g1 = c_ws * dc * eMe + dc * S1 + c_ws * S2 + S3;
      g2 = dc * dc * eMe + 2.f * dc * S2 + S4;
    }
    STAGE(11);
    float alpha = 0.f, lo = 0.f, hi = -1.f;
    for (int ls = 0; ls < 30; ++ls) {
      float d1 = s1 + g1, d2 = s2 + g2;
      if (ls > 0) {
        const float x = fmaf(alpha, jv, jar);
        const bool neg = on && x < 0.f;
        d1 = wave_sum(neg ? D * x * jv : 0.f) + g1 + alpha * g2;
        d2 = wave_sum(neg ? D * jv * jv : 0.f) + g2;
      }
      if (d2 <= 0.f || d1 == 0.f) break;
      if (d1 < 0.f) lo = alpha; else hi = alpha;
      float next = alpha - d1 / d2;
      bool bisected = false;
      if (hi >= 0.f && (next <= lo || next >= hi)) { next = 0.5f * (lo + hi); bisected = true; }
      const bool moved = on && ((fmaf(alpha, jv, jar) < 0.f) != (fmaf(next, jv, jar) < 0.f));
      const bool same = !bisected && !__any(moved);
      const float change = fabsf(next - alpha);
      alpha = next;
      if (same || change <= 8.f * 1.1920929e-07f * fabsf(next)) break;
    }
    STAGE(12);
#ifdef NMF_DUAL_DEBUG
    if (blockIdx.x == 0 && lane == 0) printf("   line search: alpha %g g1 %g g2 %g s1 %g s2 %g\n", alpha, g1, g2, s1, s2);
#endif
    if (alpha <= 0.f) {
      if (guessed) { guessed = false; mask = __ballot(on && jar < 0.f); continue; }
      // No descent the line search can measure.  In float32 the slope at alpha = 0 is a difference of sums of force x
      // acceleration products, reliable to ~1e-6 of the cost — and a cost that flat still leaves the accelerations of light
      // distal dofs open by per cent (round 4: 2.7e-2 of max |qacc| on a config-5 state where the oracle's 5th iteration was
      // this solve's missing one).  The Newton target itself does not depend on the cost: go towards it as far as the first
      // row that changes sign (the active-set step; a hair beyond it, so that the row's new state is what the next
      // elimination sees).  At most three such steps per solve.
      if (++stalls > 3) { exit_bit = kExitStall; break; }
      const bool flips = on && ((jar < 0.f) != (jar_t < 0.f)) && jv != 0.f;
      const float ai = flips ? -jar / jv : 1.f;
      alpha = fminf(1.f, wave_min(ai) * 1.001f + 1e-6f);
    }
    alpha = fminf(alpha, 4.f);           // (a minimiser far beyond the target is a quotient of rounding errors)
    const bool was_guess = guessed;      // a step towards a guessed set's target is not a Newton step: its size says nothing about convergence
    guessed = false;
    lam = fmaf(alpha, dlam, lam); c_ws = fmaf(alpha, dc, c_ws); jar = fmaf(alpha, jv, jar);
    const float newccost = wave_sum(on && jar < 0.f ? 0.5f * D * jar * jar : 0.f);
    const float dgauss = alpha * (g1 + 0.5f * alpha * g2);
    const float improvement = (ccost - newccost) - dgauss;
    gauss += dgauss; ccost = newccost;
    STAGE(13);
    // The regular end of this loop is the KKT test above (exact) or a line search that finds no descent.  MuJoCo's rule — stop
    // when an iteration improves the cost by less than the tolerance — is a guard here, from the sixth elimination on (ties: a
    // row whose residual rounds to either sign), together with the float32 rounding floor of the cost: earlier it ends solves
    // that are still moving.  A step the line search cuts short at a row's sign change improves the cost by next to nothing —
    // in float32 by less than the cost's rounding — yet the very next elimination, with that row's new state, may go all the
    // way (round 4: the worst one-step errors, up to 4e-2 of max |qacc| under 20x adhesion, were such exits).
#ifndef NMF_DUAL_EXIT_FROM
#define NMF_DUAL_EXIT_FROM 5
#endif
    if (!was_guess && iter >= NMF_DUAL_EXIT_FROM &&
        (scale * improvement < m.tolerance || improvement <= kNoiseFactor * 1.1920929e-07f * fabsf(gauss + ccost))) { exit_bit = kExitCost; break; }
    mask = __ballot(on && jar < 0.f);
  }
  // the report: how it ended, the most pivots an elimination had, what the last target violates
  resid = 0.f;
  if (exit_bit != kExitKkt) {
    const bool off = on && ((viol_last >> lane) & 1ull);
    const float viol = wave_max(off ? fabsf(jar_last) : 0.f), all = wave_max(on ? fabsf(jar_last) : 0.f);
    resid = all > 0.f ? viol * __builtin_amdgcn_rcpf(all) : 0.f;
  }
  report = kExitDual | exit_bit | (npiv > kDualRegPivots ? kExitBigPivots : 0u) | ((unsigned int)npiv << 20);
  // An end that is not the exact one and whose last target violates more than kDualResidMax of the residuals is not taken: the
  // step goes to the primal loop (physics_forward).
#ifndef NMF_NO_FALLBACK
  if (resid > kDualResidMax && !(m.solver_flags & 4)) {
    dual_restore(s, m, lane);
    report = kExitFallback | (report & kExitBigPivots) | ((unsigned int)npiv << 20);
    return -1;
  }
#endif
  // the rows' forces
  float frow = on && jar < 0.f ? -D * jar : 0.f;
  // CPU flavour: the main solver's end point (multipliers and the warm-start scalar) — the next step's warm start is expanded
  // from it, the step's acceleration from the noslip pass's forces (MuJoCo saves qacc_warmstart before its noslip solver,
  // mj_fwdConstraint; oracle/nmf_oracle.c::step does the same; so does the primal path, physics_forward)
  const float lam_main = lam, c_main = c_ws;
  // (not on the kernels whose leg factors live in HBM — ALL_POSSIBLE: a loop around the expansion keeps its 24 hinges' registers live
  // across the back edge, 17 -> 175 spilled registers; their CPU flavour takes the primal loop and noslip_primal, physics_forward)
  const bool two_pass = !kDualGlob<TP> && m.noslip_iter > 0;
  // ---- noslip post-pass (the CPU flavour's option/noslip_iterations, reference mujoco_globals.yaml:15 under mujoco.mj_step,
  // src/flygym/simulation.py:74-76; restated from MuJoCo's documentation in oracle/nmf_oracle.c::noslip): Gauss-Seidel over
  // the pairs of opposing pyramid edges with the regulariser removed — a pair (mid + y, mid - y) keeps its sum, y in
  // [-mid, mid] minimises 1/2 f^T A f + f^T j0; an update that raises the cost is undone; up to noslip_iter sweeps.
  // A's columns come out of G (DualCol), a pair's residual is two wave sums.  Not a throughput path: the batched class strips the option.
  if (two_pass) {       // (its own function: inlined, the pass's code sat in every batched step's way — 1.0 % of the headline, round 6)
    if constexpr (kDualS<TP>) frow = dual_noslip_cold<TP>(s, m, lane, ncon, frow, j0, R, smu);
    else frow = dual_noslip<TP>(s, m, lane, ncon, frow, j0, R, smu);
    lam = frow; c_ws = 0.f;       // qacc = M^-1 (qfrc_smooth + J^T f): first pass of the expansion below
  }
  // the final active set, for the next step
  {
    const unsigned long long fin = __ballot(on && jar < 0.f);
    const unsigned int nib = (unsigned int)(fin >> (4 * cc)) & 0xfu;
    if constexpr (kDualS<TP>) {
      if (on && k == 0 && hist_slot < 4) atomicOr(&s.act_hist[hist_g >> 1], nib << ((hist_g & 1) * 16 + 4 * hist_slot));
    } else {     // entry of contact cc in half (cc & 1) of word cc / 2: lane 8 i packs contacts 2 i and 2 i + 1
      const unsigned int entry = on && k == 0 && hist_slot < 4 ? 0x4000u | (unsigned int)hist_g | ((unsigned int)hist_slot << 8) | (nib << 10) : 0u;
      const unsigned int other = (unsigned int)__builtin_amdgcn_ds_bpermute(((lane + 4) & 63) << 2, (int)entry);
      if (on && (lane & 7) == 0 && (lane >> 3) < kHistLds<TP>) s.act_hist[lane >> 3] = entry | (other << 16);
    }
  }
  STAGE(9);

  // ---- qacc = qacc_smooth + c e + M^-1 J^T lambda: the rows' responses summed per hinge (root axes: wave sums; leg hinges:
  // LDS adds, the rows of a leg are few), then root-to-leaf over the factors
  float* const acc = dual_acc(s);                // [NLEG * NDL leg hinges | 6 root axes]
  // Leg hinges: the direction lanes put their products into G's LDS (dead from here on: the elimination and the noslip pass
  // are over) and lane (leg, hinge) adds those of its leg's contacts — contacts are sorted by body, a leg's are one range —
  // in contact order.  (Rounds 4-5a: one LDS float atomic per hinge and direction lane, eleven instructions that serialise on
  // the lanes of a leg: 1.1 k cycles, and a pass to clear the sums first.)  Skeletons whose products do not fit G keep the atomics.
  constexpr bool kProdInG = 64 * NDL <= dual_g_floats(NC);
  SUB_RESET();
  // One pass on the batched flavour.  CPU flavour (noslip on): two — first the noslip forces' acceleration, which is the step's
  // qacc (an output: straight to HBM on a launch's last step, qacc_out), then the main solver's, which stays in s.qacc as the
  // next step's warm start.
  for (int pass = 0, npass = two_pass ? 2 : 1; pass < npass; ++pass) {
  const float lam_x = pass ? lam_main : lam, c_x = pass ? c_main : c_ws;
  if constexpr (!kProdInG) {
    for (int i = lane; i < NLEG * NDL; i += kWave) acc[i] = 0.f;
    WSYNC();
  }
  SUB(36);
  {
    // the rows' multipliers as forces along the contact's directions (lane k < 3 of a quad: n, t1, t2): the response vectors
    // in registers are the directions'
    const float l0 = on ? lam_x : 0.f;
    const float q0 = NMF_DPP(l0, 0x00), q1 = NMF_DPP(l0, 0x55), q2 = NMF_DPP(l0, 0xAA), q3 = NMF_DPP(l0, 0xFF);
    const float fdir = k == 0 ? (q0 + q1) + (q2 + q3) : k == 1 ? mu * (q0 - q1) : k == 2 ? mu * (q2 - q3) : 0.f;
    if constexpr (kProdInG) {
      WSYNC();      // (G's last readers are done)
      if (on && k < 3) {
#pragma unroll
        for (int d = 0; d < NDL; ++d) Gm[lane * NDL + d] = fdir * ul[d];
      }
      WSYNC();
      for (int i = lane; i < NLEG * NDL; i += kWave) {
        const int g = i / NDL, d = i - g * NDL;
        const int c0 = (int)s.body_cstart[TP::LB0 + g * TP::NBL], c1 = (int)s.body_cstart[TP::LB0 + (g + 1) * TP::NBL];
        float sum = 0.f;
        for (int c = c0; c < c1; ++c) sum += (Gm[(4 * c) * NDL + d] + Gm[(4 * c + 1) * NDL + d]) + Gm[(4 * c + 2) * NDL + d];
        acc[i] = sum;
      }
    } else if (fdir != 0.f && leg >= 0) {
#pragma unroll
      for (int d = 0; d < NDL; ++d) if (d <= dlast) (void)__hip_atomic_fetch_add(&acc[leg * NDL + d], fdir * ul[d], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    SUB(37);
    float rsum[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) rsum[i] = wave_sum(fdir * ur[i]);
    if (lane < 6) {
      float v = rsum[0];
#pragma unroll
      for (int i = 1; i < 6; ++i) v = lane == i ? rsum[i] : v;
      acc[NLEG * NDL + lane] = v;
    }
  }
  WSYNC();
  SUB(38);
  {
    // (everything the two dependent chains below read is fetched first: with the reads inside the chain every step waited for
    // its own LDS round trip — 170-200 cycles per hinge against the 60 of the group sum it is made of)
    const LaneRole L = lane_role<TP>(lane);
    const int jb = TP::LD0 + L.lg * NDL;
    float rf6[6], rfr[6], racc[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) { rf6[i] = DFroot[i][6]; rfr[i] = L.mask * DFroot[i][L.rr]; racc[i] = acc[NLEG * NDL + i]; }
    float lf6[NDL], lfr[NDL], lacc[NDL], lS[NDL], lbase[NDL];
#pragma unroll
    for (int d = 0; d < NDL; ++d) {
      if constexpr (kDualGlob<TP>) { const gptr<float> q = G((const float*)DFleg) + (L.lg * NDL + d) * 8; lf6[d] = q[6]; lfr[d] = L.mask * q[L.rr]; }
      else { lf6[d] = DFleg[L.lg * NDL + d][6]; lfr[d] = L.mask * DFleg[L.lg * NDL + d][L.rr]; }
      lacc[d] = acc[L.lg * NDL + d]; lS[d] = s.S[jb + d][L.rr];
      lbase[d] = s.qacc_smooth[jb + d] + (kWarm ? c_x * s.vA[jb + d] : 0.f);
    }
    float a = 0.f, xw[6];
    static_for<6>([&](auto II) {
      constexpr int i = 5 - decltype(II)::value, e = dual_root_axis(i);
      const float xe = rf6[i] * (racc[i] - grp8_sum(rfr[i] * a));
      xw[e] = xe;
      a = L.rr == e ? a + xe : a;
    });
    if (lane < 3) {
      s.qacc[lane] = s.qacc_smooth[lane] + (kWarm ? c_x * s.vA[lane] : 0.f) + (lane == 0 ? xw[3] : lane == 1 ? xw[4] : xw[5]);
      s.qacc[3 + lane] = s.qacc_smooth[3 + lane] + (kWarm ? c_x * s.vA[3 + lane] : 0.f) + s.S[3 + lane][0] * xw[0] + s.S[3 + lane][1] * xw[1] + s.S[3 + lane][2] * xw[2];
    }
    SUB(39);
    static_for<NDL>([&](auto DD) {
      constexpr int d = decltype(DD)::value;
      const float xj = lf6[d] * (lacc[d] - grp8_sum(lfr[d] * a));
      s.qacc[jb + d] = lbase[d] + xj;
      a = fmaf(xj, lS[d], a);
    });
    SUB(40);
    if constexpr (kDualH<TP>) {
      // the rest of the body follows the root: its accelerations are the unconstrained ones + the response of the smooth
      // solve's cached factors to the root's change (one root-to-leaf pass, as at the end of the primal loop's reduced problem)
      WSYNC();       // (A, in T..W, is dead)
      if (lane < 6) s.T[0][lane] = lane == 0 ? xw[0] : lane == 1 ? xw[1] : lane == 2 ? xw[2] : lane == 3 ? xw[3] : lane == 4 ? xw[4] : xw[5];
      WSYNC();
      if (m.rest_fast) rest_levels<TP, true, false>(s, lane, [&](const auto& nd) { rest_aba_expand<TP, 3, true>(s, nd, s.qacc, L); });
      else rest_levels<TP, false, false>(s, lane, [&](const auto& nd) { rest_aba_expand<TP, 0, true>(s, nd, s.qacc, L); });
    }
  }
  WSYNC();
  if (two_pass && pass == 0) {
    if (qacc_out) { for (int j = lane; j < TP::NV; j += kWave) qacc_out[opaque(j)] = s.qacc[j]; }
    WSYNC();
  }
  }             // the factors are dead: c_w takes the contact wrenches again
  STAGE(14);
  // ---- contact wrenches and J^T f
  {
    const V3 F = v3(quad_sum(frow * drow.x), quad_sum(frow * drow.y), quad_sum(frow * drow.z));
    if (on && k == 0) stsv(s.c_w[cc], SV{cross(r, F), F});
  }
  WSYNC();
  return iters;
}

}  // namespace nmf
