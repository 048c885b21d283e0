// nmf_tree.h — sweeps of the stepping kernel for a general kinematic tree (any JointPreset: ALL_BIOLOGICAL nv = 132,
// ALL_POSSIBLE nv = 210, custom skeletons), included by nmf_step.hip.
//
// The chain-star kernels (Topo<...>: identical leg chains hanging off the root) unroll their sweeps at compile time with
// (leg, row) lanes.  A general tree has no such regularity, so here every sweep walks the tree LEVEL BY LEVEL (bodies in
// breadth-first order, tables built by the host): lane = one body of the current level, which does its whole 6-vector /
// 6x6 work alone in registers; a down sweep reads its parent's result from LDS, an up sweep its children's.  The fly's
// tree is at most 9 levels deep and 12 bodies wide, so a sweep is ~9 short wave passes.  Same algorithms, same LDS-
// resident state, same shared stages (collision, contact rows, Newton loop, sensors) as the star kernels; the oracle
// (oracle/nmf_oracle.c) is the same for both.  Slower per fly (few lanes busy per pass) — this is the generality path.
#pragma once

namespace nmf {

// f(body) for every body of levels 1.. (root excluded), a wave barrier after each level
template <class S, class F>
__device__ __forceinline__ void tree_down(const S& s, int lane, F&& f) {
  for (int lvl = 1; lvl < (int)s.t_nlevel; ++lvl) {
    const int k = (int)s.t_lvl[lvl] + lane;
    if (k < (int)s.t_lvl[lvl + 1]) f((int)s.t_body[k]);
    WSYNC();
  }
}
template <class S, class F>
__device__ __forceinline__ void tree_up(const S& s, int lane, F&& f) {
  for (int lvl = (int)s.t_nlevel - 1; lvl >= 1; --lvl) {
    const int k = (int)s.t_lvl[lvl] + lane;
    if (k < (int)s.t_lvl[lvl + 1]) f((int)s.t_body[k]);
    WSYNC();
  }
}

// Hybrid kernels with DevModel::rest_fast (three dofs per rest body, <= kRestLevels levels of <= 8 bodies): the level
// passes below fetch (body, parent, first dof, children) of all their levels as one packed word pair per lane up front
// (t_pack) instead of walking the byte tables level by level, and unroll the three dofs — the passes are chains of
// dependent LDS round trips, and this removes most of them.  Lane g < 8 takes the g-th body of a level.
struct RestNode { int b, parent, adr, ccount, cstart, k; };   // body, its parent, first dof, children (count, first slot), own slot
__device__ __forceinline__ RestNode rest_unpack(unsigned int w0, unsigned int w1) {
  RestNode nd;
  nd.b = w0 & 255; nd.parent = (w0 >> 8) & 255; nd.adr = (w0 >> 16) & 255; nd.ccount = w0 >> 24;
  nd.cstart = w1 & 255; nd.k = (w1 >> 8) & 255;
  return nd;
}
template <class TP, bool UP, class F>
__device__ __forceinline__ void rest_levels_lane(FlyLds<TP>& s, int lane, F&& f) {
  const int nl = __builtin_amdgcn_readfirstlane((int)s.t_nlevel) - 1;      // levels below the root
  unsigned int w0[kRestLevels], w1[kRestLevels];
#pragma unroll
  for (int l = 0; l < kRestLevels; ++l) { w0[l] = s.t_pack[l][lane & 7][0]; w1[l] = s.t_pack[l][lane & 7][1]; }
  static_for<kRestLevels>([&](auto I) {
    constexpr int l = UP ? kRestLevels - 1 - decltype(I)::value : decltype(I)::value;
    if (l < nl) {
      if (lane < 8 && w0[l] != 0xffffffffu) f(rest_unpack(w0[l], w1[l]));
      WSYNC();
    }
  });
}

// rigid transforms down the tree: R_b = R_parent Rrel_b, p_b = p_parent + R_parent off_b  (relm[b] = Rrel (9), off (3))
template <class TP>
__device__ void tree_kinematics_chain(FlyLds<TP>& s, const GModel& m, int lane, float (*relm)[12]) {
  auto body = [&](int b, int p) {
    const float* R = s.xmat()[p];
    const float* M = relm[b];
    st3(s.xpos()[b], ld3(s.xpos()[p]) + mat_vec(R, ld3(M + 9)));
    float out[9];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) out[3 * i + j] = R[3 * i] * M[j] + R[3 * i + 1] * M[3 + j] + R[3 * i + 2] * M[6 + j];
#pragma unroll
    for (int i = 0; i < 9; ++i) s.xmat()[b][i] = out[i];
  };
  if constexpr (TP::kStar) {
    if (m.rest_fast) { rest_levels_lane<TP, false>(s, lane, [&](const RestNode& nd) { body(nd.b, nd.parent); }); return; }
  }
  tree_down(s, lane, [&](int b) { body(b, (int)s.t_parent[b]); });
}

// body velocities -> W, bias accelerations (parent acceleration of the root = -gravity) -> T
template <class TP>
__device__ void tree_velocity_bias(FlyLds<TP>& s, const GModel& m, int lane) {
  if (lane == 0) {
    SV vt = SV{v3(0, 0, 0), v3(0, 0, 0)};
    for (int j = 0; j < 3; ++j) vt = vt + s.qvel[j] * ldsv(s.S[j]);
    SV v = vt, a = SV{v3(0, 0, 0), v3(-m.gravity[0], -m.gravity[1], -m.gravity[2])};
    for (int j = 3; j < 6; ++j) {       // the three rotational dofs of the free joint share the velocity before the joint
      a = a + s.qvel[j] * cross_motion(vt, ldsv(s.S[j]));
      v = v + s.qvel[j] * ldsv(s.S[j]);
    }
    stsv(s.W[0], v); stsv(s.T[0], a);
  }
  WSYNC();
  tree_velocity_bias_levels(s, m, lane);
}

// the levels below the root (W[0], T[0] given)
template <class TP>
__device__ void tree_velocity_bias_levels(FlyLds<TP>& s, const GModel& m, int lane) {
  if constexpr (TP::kStar) {
    if (m.rest_fast) {
      rest_levels_lane<TP, false>(s, lane, [&](const RestNode& nd) {
        SV v = ldsv(s.W[nd.parent]), a = ldsv(s.T[nd.parent]);
        SV S[3]; float qd[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) { S[d] = ldsv(s.S[nd.adr + d]); qd[d] = s.qvel[nd.adr + d]; }
#pragma unroll
        for (int d = 0; d < 3; ++d) { a = a + qd[d] * cross_motion(v, S[d]); v = v + qd[d] * S[d]; }
        stsv(s.W[nd.b], v); stsv(s.T[nd.b], a);
      });
      return;
    }
  }
  tree_down(s, lane, [&](int b) {
    const int p = (int)s.t_parent[b], adr = (int)s.t_dofadr[b], num = (int)s.t_dofnum[b];
    SV v = ldsv(s.W[p]), a = ldsv(s.T[p]);
    for (int j = adr; j < adr + num; ++j) {
      const SV S = ldsv(s.S[j]);
      a = a + s.qvel[j] * cross_motion(v, S);
      v = v + s.qvel[j] * S;
    }
    stsv(s.W[b], v); stsv(s.T[b], a);
  });
}

// T[b] = twist of body b under the generalized vector x
template <class TP>
__device__ void tree_sweep_twists(FlyLds<TP>& s, const float* x, float (*T)[row_width_tw<TP>()], const GModel& m, int lane) {
  if (lane == 0) {
    SV t = SV{v3(0, 0, 0), v3(0, 0, 0)};
    for (int j = 0; j < 6; ++j) t = t + x[j] * ldsv(s.S[j]);
    stsv(T[0], t);
  }
  WSYNC();
  tree_sweep_twists_levels(s, x, T, m, lane);
}

template <class TP>
__device__ void tree_sweep_twists_levels(FlyLds<TP>& s, const float* x, float (*T)[row_width_tw<TP>()], const GModel& m, int lane) {
  tree_down(s, lane, [&](int b) {
    const int adr = (int)s.t_dofadr[b], num = (int)s.t_dofnum[b];
    SV t = ldsv(T[(int)s.t_parent[b]]);
    for (int j = adr; j < adr + num; ++j) t = t + x[j] * ldsv(s.S[j]);
    stsv(T[b], t);
  });
}

// W[b] <- extra(b, W[b]) + sum of the children's W, for the bodies below the root, deepest level first
template <class TP, class Extra>
__device__ __forceinline__ void tree_gather_levels(FlyLds<TP>& s, float (*W)[row_width_tw<TP>()], const GModel& m, int lane, Extra&& extra) {
  tree_up(s, lane, [&](int b) {
    SV w = extra(b, ldsv(W[b]));
    const int c0 = (int)s.t_cstart[b], c1 = c0 + (int)s.t_ccount[b];
    for (int k = c0; k < c1; ++k) w = w + ldsv(W[(int)s.t_body[k]]);
    stsv(W[b], w);
  });
}

// W[b] <- sum over the subtree of b (in place; `extra(b)` adds a per-body term first), then emit(j, S_j . W[body(j)])
template <class TP, class Extra, class Emit>
__device__ __forceinline__ void tree_sweep_project(FlyLds<TP>& s, float (*W)[row_width_tw<TP>()], const GModel& m, int lane, Extra&& extra, Emit&& emit) {
  tree_gather_levels(s, W, m, lane, extra);
  if (lane == 0) {
    SV w = extra(0, ldsv(W[0]));
    const int c0 = (int)s.t_cstart[0], c1 = c0 + (int)s.t_ccount[0];
    for (int k = c0; k < c1; ++k) w = w + ldsv(W[(int)s.t_body[k]]);
    stsv(W[0], w);
  }
  WSYNC();
  for (int j = lane; j < s.nv(); j += kWave) emit(j, dot(ldsv(s.S[j]), ldsv(W[(int)s.t_dofbody[j]])));
  WSYNC();
}

// row k of the contact frame at contact c:  l = [r x d; d]
__device__ __forceinline__ void contact_dirs(V3 r, const Frame& fr, float* ln, float* l1, float* l2) {
  const V3 xn = cross(r, fr.n), x1 = cross(r, fr.t1), x2 = cross(r, fr.t2);
  ln[0] = xn.x; ln[1] = xn.y; ln[2] = xn.z; ln[3] = fr.n.x; ln[4] = fr.n.y; ln[5] = fr.n.z;
  l1[0] = x1.x; l1[1] = x1.y; l1[2] = x1.z; l1[3] = fr.t1.x; l1[4] = fr.t1.y; l1[5] = fr.t1.z;
  l2[0] = x2.x; l2[1] = x2.y; l2[2] = x2.z; l2[3] = fr.t2.x; l2[4] = fr.t2.y; l2[5] = fr.t2.z;
}

// Articulated-body solve of (CRBA(I_b [+ K_b]) + diag(delta)) x = tau on the tree; leaves T = twists(x).
//   up   : IA_b = I_b [+ contact stiffness] + children; per dof (last to first): U = IA s, D = s.U + delta,
//          u = tau - s.pA, IA -= U UT / D, pA += U u / D; (U, u, 1/D) parked in LDS; (IA, pA) handed to the parent
//   down : x_j = (u_j - U_j . a) / D_j,  a += s_j x_j
// The per-body pieces are separate so that the hybrid kernels (legs unrolled, the rest of the body as a tree) can run
// them on the rest bodies only.  TP::kFact0 = first dof that has a `fact` slot, TP::kSlot0 = first body with a `slot`.
template <class TP, bool WELD>
__device__ __forceinline__ void tree_aba_eliminate_body(FlyLds<TP>& s, int b, const float* tau, bool withK, float hdamp,
                                                        const GModel& m, const Frame& fr) {
  Sym6 IA;
  sym6_zero(IA);
  sym6_add_inertia(IA, s.Ib[b]);
  float pA[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const int c0 = (int)s.t_cstart[b], c1 = c0 + (int)s.t_ccount[b];
  for (int k = c0; k < c1; ++k) {
    const float* sl = s.slot_at((int)s.t_body[k] - TP::kSlot0);
#pragma unroll
    for (int i = 0; i < 21; ++i) IA.v[i] += sl[i];
#pragma unroll
    for (int i = 0; i < 6; ++i) pA[i] += sl[21 + i];
  }
  if (withK) {
    for (int c = s.body_cstart[b]; c < s.body_cstart[b + 1]; ++c) {
      const int act = info_act(s.c_info[c]);
      if (!act) continue;
      float ln[6], l1[6], l2[6];
      contact_dirs(ld3(s.c_r[c]), TP::kTerrain ? contact_frame(info_fid(s.c_info[c]), fr) : fr, ln, l1, l2);     // a terrain side face has its own frame
      const float D = s.c_D[c], mu = s.c_mu[c];
      const float a0 = (act & 1) ? 1.f : 0.f, a1 = (act & 2) ? 1.f : 0.f, a2 = (act & 4) ? 1.f : 0.f, a3 = (act & 8) ? 1.f : 0.f;
      // sum_k a_k D (ln +- mu lt)(ln +- mu lt)T
      sym6_rank1(IA, ln, D * (a0 + a1 + a2 + a3));
      sym6_rank1(IA, l1, D * mu * mu * (a0 + a1));
      sym6_rank1(IA, l2, D * mu * mu * (a2 + a3));
      sym6_rank2(IA, ln, l1, D * mu * (a0 - a1));
      sym6_rank2(IA, ln, l2, D * mu * (a2 - a3));
    }
    if constexpr (WELD) {
      if (b == 0) {
#pragma unroll
        for (int i = 0; i < 6; ++i) IA.v[sym_idx(i, i)] += s.weldD[i];
      }
    }
  }
  const int adr = (int)s.t_dofadr[b], num = (int)s.t_dofnum[b];
  for (int j = adr + num - 1; j >= adr; --j) {
    float sj[6], U[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) sj[i] = s.S[j][i];
    sym6_mul(IA, sj, U);
    float D = j < 6 ? 0.f : dof_delta(s, m, j, hdamp), sp = 0.f;
#pragma unroll
    for (int i = 0; i < 6; ++i) { D += sj[i] * U[i]; sp += sj[i] * pA[i]; }
    const float invD = __builtin_amdgcn_rcpf(D), u = tau[j] - sp;
    float* f = s.fact[j - TP::kFact0];
#pragma unroll
    for (int i = 0; i < 6; ++i) f[i] = U[i];
    f[6] = u; f[7] = invD;
    sym6_rank1(IA, U, -invD);
    const float ku = u * invD;
#pragma unroll
    for (int i = 0; i < 6; ++i) pA[i] += U[i] * ku;
  }
  if (b >= TP::kSlot0) {
    float* sl = s.slot_at(b - TP::kSlot0);
#pragma unroll
    for (int i = 0; i < 21; ++i) sl[i] = IA.v[i];
#pragma unroll
    for (int i = 0; i < 6; ++i) sl[21 + i] = pA[i];
  }
}

template <class TP>
__device__ __forceinline__ void tree_aba_expand_body(FlyLds<TP>& s, int b, SV a, float* x, const GModel& m) {
  const int adr = (int)s.t_dofadr[b], num = (int)s.t_dofnum[b];
  for (int j = adr; j < adr + num; ++j) {
    const float* f = s.fact[j - TP::kFact0];
    const SV U = ldsv(f), S = ldsv(s.S[j]);
    const float xj = (f[6] - dot(U, a)) * f[7];
    x[j] = xj;
    a = a + xj * S;
  }
  stsv(s.T[b], a);
}

// ---------------------------------------------------------------- hybrid kernels: the rest's ABA in the leg chains' lane layout
// One 8-lane group per body of a level (the fly's rest has at most 6 bodies per level): lane r < 6 of the group carries
// row r of the body's articulated inertia and component r of its bias wrench, group sums by DPP.  Same LDS hand-off as
// above (s.fact per dof, s.slot: upper triangle + bias wrench), the slots in breadth-first order so that the children
// of a body are contiguous.
// A level pass is a chain of dependent LDS round trips (level table -> body -> dof range -> axes -> ...), ~100 cycles
// each with one or two waves per SIMD, and that chain, not the arithmetic, is its cost.  FAST (DevModel::rest_fast: three
// dofs per body, <= kRestLevels levels of <= 8 bodies): the (level, group) -> body / parent / dofs / children table is
// one packed 8-byte word per lane and level, fetched for all levels at once, and the dof loops are unrolled, so every
// load of a body is issued up front — one round trip per body plus one per child.
template <class TP>
__device__ __forceinline__ RestNode rest_node_tbl(const FlyLds<TP>& s, int k) {
  RestNode nd;
  nd.k = k; nd.b = (int)s.t_body[k];
  nd.parent = (int)s.t_parent[nd.b]; nd.adr = (int)s.t_dofadr[nd.b];
  nd.ccount = (int)s.t_ccount[nd.b]; nd.cstart = (int)s.t_cstart[nd.b];
  return nd;
}

// f(node) for every body of the rest, level by level: UP = deepest level first
template <class TP, bool FAST, bool UP, class F>
__device__ __forceinline__ void rest_levels(FlyLds<TP>& s, int lane, F&& f) {
  const int nl = __builtin_amdgcn_readfirstlane((int)s.t_nlevel) - 1;      // levels below the root
  if constexpr (FAST) {
    unsigned int w0[kRestLevels], w1[kRestLevels];
#pragma unroll
    for (int l = 0; l < kRestLevels; ++l) { w0[l] = s.t_pack[l][lane >> 3][0]; w1[l] = s.t_pack[l][lane >> 3][1]; }
    static_for<kRestLevels>([&](auto I) {
      constexpr int l = UP ? kRestLevels - 1 - decltype(I)::value : decltype(I)::value;
      if (l < nl) {
        if (w0[l] != 0xffffffffu) f(rest_unpack(w0[l], w1[l]));
        WSYNC();
      }
    });
  } else {
    for (int i = 0; i < nl; ++i) {
      const int lvl = UP ? nl - i : 1 + i;
      for (int k = (int)s.t_lvl[lvl] + (lane >> 3); k < (int)s.t_lvl[lvl + 1]; k += 8) f(rest_node_tbl(s, k));
      WSYNC();
    }
  }
}

// the dofs of a body, last to first (REV) or first to last; NUM > 0: a compile-time count (fully unrolled, so the LDS
// loads of all dofs are issued up front), NUM == 0: the run-time count
template <int NUM, bool REV, class F>
__device__ __forceinline__ void rest_dofs(int adr, int num, F&& f) {
  if constexpr (NUM > 0) {
    static_for<NUM>([&](auto I) { constexpr int i = decltype(I)::value; f(adr + (REV ? NUM - 1 - i : i)); });
  } else if (REV) { for (int j = adr + num - 1; j >= adr; --j) f(j); }
  else { for (int j = adr; j < adr + num; ++j) f(j); }
}

// full elimination of a body: matrix factors and vector part
template <class TP, int NUM>
__device__ __forceinline__ void rest_aba_eliminate(FlyLds<TP>& s, const RestNode& nd, const float* tau, bool withK, float hdamp,
                                                   const Frame& fr, const LaneRole& L, const int (&so)[6], const InertiaRowMap& IM, const GModel& m) {
  const int num = NUM > 0 ? NUM : (int)s.t_dofnum[nd.b];
  // everything that depends on the node only, first: the body's inertia row and its dofs' axes and scalars
  float row[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  add_inertia_row(row, s, nd.b, IM);
  float sj[NUM > 0 ? NUM : 1][6], sown[NUM > 0 ? NUM : 1], delta[NUM > 0 ? NUM : 1], tj[NUM > 0 ? NUM : 1];
  if constexpr (NUM > 0) {
#pragma unroll
    for (int d = 0; d < NUM; ++d) {
      const int j = nd.adr + d;
#pragma unroll
      for (int i = 0; i < 6; ++i) sj[d][i] = s.S[j][i];
      sown[d] = s.S[j][L.rr]; delta[d] = dof_delta(s, m, j, hdamp); tj[d] = tau[j];
    }
  }
  float pA = 0.f;
  for (int k = nd.cstart - 1; k < nd.cstart - 1 + nd.ccount; ++k) {
    const float* sl = s.slot_at(k);
    { float v[6];
#pragma unroll
      for (int c = 0; c < 6; ++c) v[c] = sl[so[c]];
      add6(row, v); }
    pA += sl[21 + L.rr];
  }
  if (withK) {
    KLane KL;
    const float* q = s.k_tab[L.rr];
#pragma unroll
    for (int i = 0; i < 3; ++i) { KL.dA[i] = q[i]; KL.dB[i] = q[3 + i]; KL.dO[i] = q[6 + i]; }
    KL.ia = __float_as_int(q[9]); KL.ib = __float_as_int(q[10]);
    const bool walls = TP::kTerrain && __builtin_amdgcn_readfirstlane(s.nwall) != 0;
    for (int c = s.body_cstart[nd.b]; c < s.body_cstart[nd.b + 1]; ++c) add_contact_K_row(row, s, c, KL, fr, L.rr, walls);
  }
  if constexpr (NUM > 0) {
    static_for<NUM>([&](auto I) {
      constexpr int d = NUM - 1 - decltype(I)::value;
      float U, u, invD;
      aba_step(row, pA, sj[d], sown[d], L.mask, delta[d], tj[d], U, u, invD);
      float* f = s.fact[nd.adr + d - TP::kFact0];
      if (L.r < 6) f[L.rr] = U;
      if (L.r == 0) { f[6] = u; f[7] = invD; }
    });
  } else {
    for (int j = nd.adr + num - 1; j >= nd.adr; --j) {
      float s1[6];
#pragma unroll
      for (int i = 0; i < 6; ++i) s1[i] = s.S[j][i];
      float U, u, invD;
      aba_step(row, pA, s1, s.S[j][L.rr], L.mask, dof_delta(s, m, j, hdamp), tau[j], U, u, invD);
      float* f = s.fact[j - TP::kFact0];
      if (L.r < 6) f[L.rr] = U;
      if (L.r == 0) { f[6] = u; f[7] = invD; }
    }
  }
  if (L.r < 6) {
    float* sl = s.slot_at(nd.k - 1);
#pragma unroll
    for (int c = 0; c < 6; ++c) if (c >= L.rr) sl[so[c]] = row[c];
    sl[21 + L.rr] = pA;
  }
}

// back-substitution of a body given its parent's acceleration in T; HOMOGENEOUS: no force on the subtree, the dofs'
// accelerations are written to qacc as changes of the unconstrained ones (reduced constraint problem, physics_forward)
template <class TP, int NUM, bool HOMOGENEOUS>
__device__ __forceinline__ void rest_aba_expand(FlyLds<TP>& s, const RestNode& nd, float* x, const LaneRole& L) {
  const int num = NUM > 0 ? NUM : (int)s.t_dofnum[nd.b];
  float a = s.T[nd.parent][L.rr];
  if constexpr (NUM > 0) {
    float sown[NUM], Uj[NUM], uj[NUM], invD[NUM], base[NUM];
#pragma unroll
    for (int d = 0; d < NUM; ++d) {
      const int j = nd.adr + d;
      const float* f = s.fact[j - TP::kFact0];
      sown[d] = s.S[j][L.rr]; Uj[d] = L.mask * f[L.rr]; uj[d] = HOMOGENEOUS ? 0.f : f[6]; invD[d] = f[7];
      base[d] = HOMOGENEOUS ? s.qacc_smooth[j] : 0.f;
    }
    static_for<NUM>([&](auto I) {
      constexpr int d = decltype(I)::value;
      const float xj = (uj[d] - grp8_sum(Uj[d] * a)) * invD[d];
      if (L.r == 0) { if (HOMOGENEOUS) s.qacc[nd.adr + d] = base[d] + xj; else x[nd.adr + d] = xj; }
      a += xj * sown[d];
    });
  } else {
    for (int j = nd.adr; j < nd.adr + num; ++j) {
      const float* f = s.fact[j - TP::kFact0];
      const float Ua = grp8_sum(L.mask * f[L.rr] * a);
      const float xj = HOMOGENEOUS ? -Ua * f[7] : (f[6] - Ua) * f[7];
      if (L.r == 0) { if (HOMOGENEOUS) s.qacc[j] = s.qacc_smooth[j] + xj; else x[j] = xj; }
      a += xj * s.S[j][L.rr];
    }
  }
  if (L.r < 6) s.T[nd.b][L.rr] = a;
}

template <class TP, bool WELD>
__device__ void tree_aba_solve(FlyLds<TP>& s, int tau_id, int x_id, bool withK, float hdamp, const GModel& m, int lane) {
  const float* tau = s.vec(tau_id);
  float* x = s.vec(x_id);
  const Frame fr = ld_frame(s, m);
  tree_up(s, lane, [&](int b) { tree_aba_eliminate_body<TP, WELD>(s, b, tau, withK, hdamp, m, fr); });
  if (lane == 0) tree_aba_eliminate_body<TP, WELD>(s, 0, tau, withK, hdamp, m, fr);
  WSYNC();
  if (lane == 0) tree_aba_expand_body(s, 0, SV{v3(0, 0, 0), v3(0, 0, 0)}, x, m);
  WSYNC();
  tree_down(s, lane, [&](int b) { tree_aba_expand_body(s, b, ldsv(s.T[(int)s.t_parent[b]]), x, m); });
}

}  // namespace nmf
