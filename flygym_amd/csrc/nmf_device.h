// nmf_device.h — device-side data structures and small math for the gfx950 stepping kernels.
//
// One wavefront (64 lanes) owns one fly for a whole launch: the fly's state and every
// intermediate quantity live in LDS; model constants are read-only and shared by all flies
// (L2 resident).  The per-step pipeline restates MuJoCo's documented mj_step for the model the
// reference builds (see oracle/nmf_oracle.c for the stage list and the reference citations).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>
#include <utility>

namespace nmf {

constexpr int kWave = 64;
constexpr int kRestLevels = 6;   // levels below the root the fast passes of the hybrid kernels unroll
constexpr int kMaxCon = 48;      // contacts per fly kept by the engine (overflow is flagged)
constexpr int kMaxCtrl = 48;
constexpr int kActHistWords = 64; // 16 bits per geom, 128 geoms
constexpr float kMinVal = 1e-15f;

enum { GEOM_CAPSULE = 0, GEOM_HULL = 1 };
enum { ACT_POSITION = 0, ACT_ADHESION = 1, ACT_MOTOR = 2 };

// Star-of-chains topology: one free root body + NLEG identical serial chains; DOFS... are the
// hinge counts of the chain's bodies from the root outwards (LEGS_ONLY leg: 3,2,1,1,1,1,1,1).
// Everything about the chain layout is a compile-time constant so that the leg sweeps unroll
// completely and never load structure from memory.
// REST_B / REST_V: bodies / dofs of the "rest" of the fly (head, antennae, proboscis, abdomen, wings, halteres) that sit
// between the root and the legs in the model's order; they are swept by the general-tree code (nmf_tree.h), the legs by
// the unrolled chain code.  Leg-only skeletons have no rest.
template <int REST_B_, int REST_V_, int NLEG_, int... DOFS>
struct HybridTopo {
  static constexpr bool kStar = true;
  static constexpr bool kTerrain = false;   // see Terrain<> below
  // controls a kernel keeps in LDS: 48 for the leg skeletons, 64 for ALL_BIOLOGICAL (which sits exactly on its LDS budget), 96 for
  // ALL_POSSIBLE (the default actuated set on it is 72 leg dofs + 6 adhesion); nmf_batch_create sends models with more
  // actuators to the general-tree kernel, which holds one per dof
  static constexpr int kCtrl = REST_V_ == 0 ? kMaxCtrl : ((DOFS + ...) > 16 ? 96 : 64);
  static constexpr int REST_B = REST_B_, REST_V = REST_V_;
  static constexpr int NLEG = NLEG_;
  static constexpr int NBL = sizeof...(DOFS);
  static constexpr int NDL = (DOFS + ...);
  static constexpr int LB0 = 1 + REST_B_;        // first leg body
  static constexpr int LD0 = 6 + REST_V_;        // first leg dof
  static constexpr int kFact0 = 6, kSlot0 = 1;   // the rest dofs / bodies only (tree sweeps); legs and root keep theirs in registers
  static constexpr int kNFact = REST_V_ > 0 ? REST_V_ : 1, kNSlot = REST_B_ > 0 ? REST_B_ : 1;
  static constexpr int kTblB = 1 + REST_B_, kTblV = 6 + REST_V_;   // tree tables: root + rest bodies, root + rest dofs
  static constexpr int NB = LB0 + NLEG_ * NBL;
  static constexpr int NV = LD0 + NLEG_ * NDL;
  static constexpr int NQ = NV + 1;
  static constexpr int dofs(int l) { constexpr int t[] = {DOFS...}; return t[l]; }
  static constexpr int first_dof(int l) { int a = 0; for (int i = 0; i < l; ++i) a += dofs(i); return a; }
  static constexpr int lbody(int d) { int a = 0; for (int l = 0; l < NBL; ++l) { a += dofs(l); if (d < a) return l; } return NBL - 1; }
  static constexpr bool is_last(int d) { return d == first_dof(lbody(d)) + dofs(lbody(d)) - 1; }
  static constexpr bool is_first(int d) { return d == first_dof(lbody(d)); }
};
template <int NLEG_, int... DOFS>
using Topo = HybridTopo<0, 0, NLEG_, DOFS...>;

// A general kinematic tree (nmf_tree.h): LDS arrays sized for NB_ bodies / NV_ dofs, the actual counts are run-time
// values of the model.  Two sizes are built: 72 x 144 (ALL_BIOLOGICAL: 69 bodies, 132 dofs; 4 flies per CU) and
// 72 x 216 (ALL_POSSIBLE: 210 dofs; 3 flies per CU).
template <int NB_, int NV_>
struct TreeTopoT {
  static constexpr bool kStar = false;
  static constexpr bool kTerrain = false;
  static constexpr int NB = NB_, NV = NV_, NQ = NV_ + 1;
  static constexpr int kCtrl = NV_ + 8;      // every dof actuated + adhesion
  static constexpr int kFact0 = 0, kSlot0 = 1;      // every dof has articulated-body factors, every non-root body a hand-off slot
  static constexpr int kNFact = NV_, kNSlot = NB_;
  static constexpr int kTblB = NB_, kTblV = NV_;
};
// The same skeleton in a world with a terrain (gapped / blocks / mixed: cells with tops and side faces).  A compile-time
// property of the kernel: the collision stage against the cells, contacts with their own frames (a side face's normal is
// horizontal) in every stage that uses the contact frame.  Flat and tethered worlds run the kernels without any of it —
// the same code, registers and LDS as before the terrain's side faces existed.
template <class TP>
struct Terrain : TP {
  static constexpr bool kTerrain = true;
};
using TreeTopo = TreeTopoT<72, 216>;
using TreeTopoSmall = TreeTopoT<72, 144>;

// The same value, but not one the optimizer can see through: address arithmetic built on it stays where it is written.
// Used for the once-per-item / once-per-launch paths of the stepping kernel (state hand-over, pure outputs): hoisted out
// of the persistent item loop, their per-lane 64-bit addresses sat in ~40 vector registers across every step and 31 of
// them were spilled to scratch (round 2's shipped kernel).
__device__ __forceinline__ int opaque(int v) { asm volatile("" : "+v"(v)); return v; }

template <int... I, class F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, I...>, F&& f) {
  (f(std::integral_constant<int, I>{}), ...);
}
// static_for<N>([&](auto I) { constexpr int i = decltype(I)::value; ... });
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(std::make_integer_sequence<int, N>{}, static_cast<F&&>(f));
}

// Device-side view of pointers into HBM: typed as global memory, so that loads through them are global_load (own counter,
// SGPR base) instead of flat_load (LDS-aperture check, waits tied to the LDS counter).  The host sees plain pointers.
#if defined(__HIP_DEVICE_COMPILE__)
#define NMF_G __attribute__((address_space(1)))
#else
#define NMF_G
#endif

struct DevModel {
  int nb, nv, nq, nu, ng, nseg, nsite, nsensor, max_iter;
  float timestep, tolerance, hull_skin, meaninertia;
  float gravity[3];
  float plane[4];
  int weld_active;          // tether: soft weld of the root body to weld_pos / weld_quat (TetheredWorld)
  float weld_pos[3], weld_quat[4], weld_solref[2], weld_solimp[5], weld_invweight[2];
  int terrain_type;         // 0 flat, 1 gapped, 2 blocks, 3 mixed
  float terrain[5];         // parameters + maximum height (see flygym_amd/compose/world.py)
  const NMF_G float *body_pos, *body_quat, *body_mass, *body_ipos, *body_inertia;
  const NMF_G int *body_dofadr, *body_dofnum, *dof_body;
  // general-tree kernel only: bodies in breadth-first order (level by level, children of a body contiguous)
  const NMF_G int *body_parent, *tree_body, *tree_child_start, *tree_child_count;   // child ranges index tree_body
  int tree_nlevel, tree_lvl_start[18];
  // hybrid kernels, fast level passes: every body of the rest has exactly three dofs, at most kRestLevels levels of at
  // most 8 bodies.  rest_pack[level - 1][group][2]: body | parent << 8 | first dof << 16 | children << 24,
  // first child (breadth-first slot) | own breadth-first slot << 8;  0xffffffff = no body for that group
  int rest_fast;
  const NMF_G unsigned int* rest_pack;
  const NMF_G float *dof_axis, *dof_armature, *dof_damping, *dof_stiffness, *dof_springref;
  const NMF_G int* seg_body;
  const NMF_G float *seg_pos, *seg_quat;
  const NMF_G int* site_body;
  const NMF_G float* site_pos;
  const NMF_G int *act_type, *act_trn, *act_limited;
  const NMF_G int* act_geom;      // adhesion actuators: contact geom of the adhesion segment (-1: none)
  // named engine semantics (blob entry sem_options; flygym_amd.compiler.model.EngineSemantics), shared with the oracle
  int sem_pyramid_plain, sem_adhesion_fused, sem_sensor_contact_frame, sem_max_hull_contacts;
  int sem_terrain_walls;    // terrains: the cells' side faces collide (flygym_amd/compose/world.py::terrain_probe)
  // diagnostics (environment NMF_SOLVER at batch creation): bit 0 = every step on the primal Newton loop, bit 1 = the contact-space
  // solve starts from the start point's own sign pattern instead of the previous step's active set.  Same optimum either way.
  int solver_flags;
  // option/noslip_iterations of the CPU flavour (reference mujoco_globals.yaml:15): sweeps of the friction-only post-pass after the
  // Newton solve (contact-space solve only, nmf_dual.h); 0 on the batched path, as the reference's GPU class sets it
  int noslip_iter;
  // contacts kept per world and step (nmf_batch_set_contact_capacity; HIPSimulation's max_contacts, reference
  // warp/simulation.py:50-56): 1..kMaxCon.  Contacts beyond it — in geom order — are dropped and the step is counted as overflowed.
  int max_contacts;
  const NMF_G float *act_gain, *act_bias, *act_forcerange, *act_ctrlrange;
  // [nu][32] or nullptr: MuJoCo's general actuator for the actuators the affine pass does not cover (intvelocity, damper,
  // cylinder, muscle; later actuators of a shared dof) — flygym_amd/compiler/model.py::_general_row, nmf_step.hip actuation_general
  const NMF_G float* act_general;
  const NMF_G float *key_qpos, *key_ctrl;
  const NMF_G int *geom_body, *geom_type, *geom_hulladr, *geom_hullnum, *geom_sensor;
  const NMF_G float *geom_p0, *geom_p1, *geom_radius, *geom_bsphere, *geom_invweight0, *hull_vert;
  const NMF_G float *pair_friction, *pair_solref, *pair_solimp, *pair_margin;
};

using GModel = NMF_G DevModel;   // the model as the device functions see it (global memory)

struct DevState {
  int n_worlds;
  float *qpos, *qvel, *ctrl, *qacc_ws, *seg_xpos, *seg_xquat, *site_xpos, *actuator_force,
      *sensordata, *time, *stats, *qacc;
  float* act;              // [n_worlds][nu] activation state of the stateful actuators (one slot per actuator, 0 for the stateless)
  unsigned int* stats_sum; // [n_worlds][16] since the last reset: physics steps, sum of contacts, sum of Newton iterations, overflow steps, 12 solve-report counters (include/nmf.h)
  float* contact_geom;     // [n_worlds][kMaxCon] geom index of contact c at the launch's last step (-1 beyond ncon)
  // [n_worlds][kActHistWords] the constraint solver's second warm start: the active pyramid rows (4 bits) of up to four contacts
  // per geom at the end of the last step the contact-space solve ran (nmf_dual.h); zero = nothing known
  unsigned int* act_hist;
  float* cost;             // [n_worlds] shader cycles world w took in the last stepping launch
  const int* order;        // [n_worlds] block -> world (nullptr: identity); scheduling only
  struct SchedState* sched;   // launch-duration bookkeeping of the block-order policy (nullptr: off)
  // chunked launches (nmf_step_kernel): a launch of n_steps is cut into chunks (chunk_start); workgroups take
  // (world, chunk) items from a ticket counter, a world's chunks hand its state over through HBM
  struct ChunkSched* csched;
  // hand-off of a world's state between two chunks of one launch: [n_worlds][handoff_stride] 8-byte granules
  // {float bits, tag}, tag = epoch * 32 + chunks of this launch the world has finished (see nmf_step_kernel)
  unsigned long long* handoff;
  int handoff_stride;
  int n_chunks;               // chunked schedule: number of chunks
  unsigned long long* clock_probe;   // [2] shader cycles / 100 MHz ticks workgroup 0 spent in stepping launches (nmf_shader_clock)
  int sched_mode;             // 0 plain (one workgroup per world), 1 chunked (persistent workgroups, tickets): nmf_step_kernel
  // observation ring of this launch (nmf_step_record; nullptr: none): after every obs_every-th step the world's observation
  // block [joint angles ring_nj | joint velocities ring_nj | forces of the first ring_nact actuators | 96 contact-sensor floats]
  // goes to ring[(step + 1) / obs_every - 1][world][.], rows ring_stride floats apart
  float* ring;
  int ring_stride, obs_every, ring_nj, ring_nact;
  // CPU flavour (noslip iterations on): per world the scratch of the primal path's noslip pass (nmf_step.hip::noslip_primal);
  // nullptr on the batched path
  float* noslip_buf;
  // kernels whose leg factors do not fit LDS (ALL_POSSIBLE): per workgroup of a stepping launch the contact-space solve's leg
  // factors (nmf_step.hip kDualGlob); nullptr elsewhere
  float* dual_scratch;
  int chunk_start[17];        // chunk c covers steps chunk_start[c] .. chunk_start[c + 1] - 1 (lengths shrink towards the end)
};

// Device-resident scheduler state of chunked launches; the last workgroup to run out of tickets rewinds it, so launches need no
// host-side counters (hipGraph replays stay valid).
struct ChunkSched { unsigned int ticket, exited, epoch, pad; };

// Device-resident state of the block-order policy (see nmf_order_kernel)
struct SchedState {
  unsigned long long t_first, t_last;   // earliest block start / latest block end of the last stepping launch (s_memrealtime)
  float ema[2];                          // smoothed launch duration per physics step, per policy (0 in-order, 1 costliest first)
  int last_policy, launches, last_steps, pad;
};

struct ReplayArgs {
  const float* table;   // [n_worlds][table_steps][n_act] or nullptr
  const int* act_ids;   // [n_act]
  int table_steps, n_act, start;
};

// ---------------------------------------------------------------- small math
struct V3 { float x, y, z; };
__device__ __forceinline__ V3 v3(float x, float y, float z) { return V3{x, y, z}; }
__device__ __forceinline__ V3 ld3(const float* p) { return V3{p[0], p[1], p[2]}; }
__device__ __forceinline__ void st3(float* p, V3 a) { p[0] = a.x; p[1] = a.y; p[2] = a.z; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 operator*(float s, V3 a) { return V3{s * a.x, s * a.y, s * a.z}; }
__device__ __forceinline__ float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3 cross(V3 a, V3 b) {
  return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
__device__ __forceinline__ V3 mat_vec(const float* m, V3 v) {
  return V3{m[0] * v.x + m[1] * v.y + m[2] * v.z, m[3] * v.x + m[4] * v.y + m[5] * v.z,
            m[6] * v.x + m[7] * v.y + m[8] * v.z};
}
__device__ __forceinline__ V3 matT_vec(const float* m, V3 v) {
  return V3{m[0] * v.x + m[3] * v.y + m[6] * v.z, m[1] * v.x + m[4] * v.y + m[7] * v.z,
            m[2] * v.x + m[5] * v.y + m[8] * v.z};
}

struct Q4 { float w, x, y, z; };
__device__ __forceinline__ Q4 ldq(const float* p) { return Q4{p[0], p[1], p[2], p[3]}; }
__device__ __forceinline__ void stq(float* p, Q4 q) { p[0] = q.w; p[1] = q.x; p[2] = q.y; p[3] = q.z; }
__device__ __forceinline__ Q4 qmul(Q4 a, Q4 b) {
  return Q4{a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z, a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
            a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x, a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w};
}
__device__ __forceinline__ Q4 qnorm(Q4 q) {
  float n = sqrtf(q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z);
  if (n < kMinVal) return Q4{1.f, 0.f, 0.f, 0.f};
  float s = 1.0f / n;
  return Q4{q.w * s, q.x * s, q.y * s, q.z * s};
}
__device__ __forceinline__ V3 qrot_conj(Q4 q, V3 v) {  // rotate v by q^-1 (q unit)
  V3 u = v3(-q.x, -q.y, -q.z);
  V3 t = 2.0f * cross(u, v);
  return v + q.w * t + cross(u, t);
}
__device__ __forceinline__ void qmat(float* m, Q4 q) {
  float w = q.w, x = q.x, y = q.y, z = q.z;
  m[0] = 1 - 2 * (y * y + z * z); m[1] = 2 * (x * y - w * z); m[2] = 2 * (x * z + w * y);
  m[3] = 2 * (x * y + w * z); m[4] = 1 - 2 * (x * x + z * z); m[5] = 2 * (y * z - w * x);
  m[6] = 2 * (x * z - w * y); m[7] = 2 * (y * z + w * x); m[8] = 1 - 2 * (x * x + y * y);
}

__device__ __forceinline__ Q4 mat_quat(const float* m) {   // rotation matrix -> unit quaternion
  const float t = m[0] + m[4] + m[8];
  Q4 q;
  if (t > 0.f) {
    const float r = sqrtf(t + 1.f) * 2.f;
    q = Q4{0.25f * r, (m[7] - m[5]) / r, (m[2] - m[6]) / r, (m[3] - m[1]) / r};
  } else if (m[0] > m[4] && m[0] > m[8]) {
    const float r = sqrtf(1.f + m[0] - m[4] - m[8]) * 2.f;
    q = Q4{(m[7] - m[5]) / r, 0.25f * r, (m[1] + m[3]) / r, (m[2] + m[6]) / r};
  } else if (m[4] > m[8]) {
    const float r = sqrtf(1.f + m[4] - m[0] - m[8]) * 2.f;
    q = Q4{(m[2] - m[6]) / r, (m[1] + m[3]) / r, 0.25f * r, (m[5] + m[7]) / r};
  } else {
    const float r = sqrtf(1.f + m[8] - m[0] - m[4]) * 2.f;
    q = Q4{(m[3] - m[1]) / r, (m[2] + m[6]) / r, (m[5] + m[7]) / r, 0.25f * r};
  }
  return q;
}

// Packed float32 arithmetic (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32: two lanes' worth of fma per instruction).  A wave
// issues one instruction per >= 4 cycles whatever it is (scripts/valu_issue_microbench: v_pk_fma_f32 4.56 cycles, as
// v_fma_f32) and this kernel keeps the vector pipe 25 % busy, so pairing halves the issue cost of the row arithmetic.
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2 mk2(float a, float b) { f2 r; r.x = a; r.y = b; return r; }
// a[0..5] += b[0..5] / a[0..5] += k * b[0..5] as three packed operations
__device__ __forceinline__ void add6(float (&a)[6], const float* b) {
#pragma unroll
  for (int i = 0; i < 3; i++) { const f2 r = mk2(a[2 * i], a[2 * i + 1]) + mk2(b[2 * i], b[2 * i + 1]); a[2 * i] = r.x; a[2 * i + 1] = r.y; }
}
__device__ __forceinline__ void fma6(float (&a)[6], float k, const float* b) {
  const f2 kk = mk2(k, k);
#pragma unroll
  for (int i = 0; i < 3; i++) { const f2 r = __builtin_elementwise_fma(kk, mk2(b[2 * i], b[2 * i + 1]), mk2(a[2 * i], a[2 * i + 1])); a[2 * i] = r.x; a[2 * i + 1] = r.y; }
}

// spatial 6-vectors: motion (w; v) and force (n; f), world axes, about the root origin
struct SV { V3 a, l; };  // angular part, linear part
__device__ __forceinline__ SV ldsv(const float* p) { return SV{ld3(p), ld3(p + 3)}; }
__device__ __forceinline__ void stsv(float* p, SV s) { st3(p, s.a); st3(p + 3, s.l); }
__device__ __forceinline__ SV operator+(SV x, SV y) { return SV{x.a + y.a, x.l + y.l}; }
__device__ __forceinline__ SV operator-(SV x, SV y) { return SV{x.a - y.a, x.l - y.l}; }
__device__ __forceinline__ SV operator*(float s, SV x) { return SV{s * x.a, s * x.l}; }
__device__ __forceinline__ float dot(SV x, SV y) { return dot(x.a, y.a) + dot(x.l, y.l); }
__device__ __forceinline__ SV cross_motion(SV a, SV b) {
  return SV{cross(a.a, b.a), cross(a.a, b.l) + cross(a.l, b.a)};
}
__device__ __forceinline__ SV cross_force(SV v, SV f) {
  return SV{cross(v.a, f.a) + cross(v.l, f.l), cross(v.a, f.l)};
}
// rigid-body spatial inertia about the reference point: m, h = m c, I sym6 (xx yy zz xy xz yz)
__device__ __forceinline__ SV inert_mul(const float* I, SV v) {
  V3 h = ld3(I + 1);
  V3 Iw = v3(I[4] * v.a.x + I[7] * v.a.y + I[8] * v.a.z, I[7] * v.a.x + I[5] * v.a.y + I[9] * v.a.z,
             I[8] * v.a.x + I[9] * v.a.y + I[6] * v.a.z);
  return SV{Iw + cross(h, v.l), I[0] * v.l - cross(h, v.a)};
}

// symmetric 6x6 in 21 floats, upper triangle row-major
__device__ __host__ constexpr int sym_idx(int i, int j) {
  return i <= j ? i * 6 - i * (i - 1) / 2 + (j - i) : j * 6 - j * (j - 1) / 2 + (i - j);
}
struct Sym6 {
  float v[21];
  __device__ __forceinline__ float get(int i, int j) const { return v[sym_idx(i, j)]; }
};
__device__ __forceinline__ void sym6_zero(Sym6& A) {
#pragma unroll
  for (int i = 0; i < 21; i++) A.v[i] = 0.f;
}
__device__ __forceinline__ void sym6_mul(const Sym6& A, const float* s, float* out) {
#pragma unroll
  for (int i = 0; i < 6; i++) {
    float acc = 0.f;
#pragma unroll
    for (int j = 0; j < 6; j++) acc += A.v[sym_idx(i, j)] * s[j];
    out[i] = acc;
  }
}
// A += c * l lᵀ
__device__ __forceinline__ void sym6_rank1(Sym6& A, const float* l, float c) {
#pragma unroll
  for (int i = 0; i < 6; i++) {
    float ci = c * l[i];
#pragma unroll
    for (int j = i; j < 6; j++) A.v[sym_idx(i, j)] += ci * l[j];
  }
}
// A += c * (l mᵀ + m lᵀ)
__device__ __forceinline__ void sym6_rank2(Sym6& A, const float* l, const float* m, float c) {
#pragma unroll
  for (int i = 0; i < 6; i++)
#pragma unroll
    for (int j = i; j < 6; j++) A.v[sym_idx(i, j)] += c * (l[i] * m[j] + m[i] * l[j]);
}
// sin and cos of a bounded angle (joint angles, |x| up to a few hundred): Cody-Waite reduction to [-pi/4, pi/4] with a
// three-part pi/2 and the single-precision minimax kernels; <= 1.5 ulp (9.2e-8 absolute, checked against double on
// [-40, 40]).  No large-argument path, and plain arithmetic: unaffected by the build's approximate-function flag.
__device__ __forceinline__ void sincos_bounded(float x, float* sn, float* cs) {
  const float k = rintf(x * 0.636619772f);
  float r = fmaf(-k, 1.57079625129699707031f, x);
  r = fmaf(-k, 7.54978941586159635335e-08f, r);
  r = fmaf(-k, 5.39030285815811905e-15f, r);
  const float z = r * r;
  const float sp = fmaf(fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f);
  const float s = fmaf(r * z, sp, r);
  const float cp = fmaf(fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f);
  const float c = fmaf(z * z, cp, fmaf(-0.5f, z, 1.0f));
  const int q = (int)k & 3;
  const float ss = (q & 1) ? c : s, cc = (q & 1) ? s : c;
  *sn = (q & 2) ? -ss : ss;
  *cs = ((q + 1) & 2) ? -cc : cc;
}

// A += rigid-body inertia (m, h, I) as a 6x6 in (w; v) ordering
__device__ __forceinline__ void sym6_add_inertia(Sym6& A, const float* I) {
  A.v[sym_idx(0, 0)] += I[4]; A.v[sym_idx(1, 1)] += I[5]; A.v[sym_idx(2, 2)] += I[6];
  A.v[sym_idx(0, 1)] += I[7]; A.v[sym_idx(0, 2)] += I[8]; A.v[sym_idx(1, 2)] += I[9];
  float hx = I[1], hy = I[2], hz = I[3];
  // block(0:3, 3:6) = [h]x
  A.v[sym_idx(0, 4)] += -hz; A.v[sym_idx(0, 5)] += hy;
  A.v[sym_idx(1, 3)] += hz;  A.v[sym_idx(1, 5)] += -hx;
  A.v[sym_idx(2, 3)] += -hy; A.v[sym_idx(2, 4)] += hx;
  A.v[sym_idx(3, 3)] += I[0]; A.v[sym_idx(4, 4)] += I[0]; A.v[sym_idx(5, 5)] += I[0];
}

// ---------------------------------------------------------------- 8-lane group reductions (DPP)
// Lanes are used as 8 groups of 8; a group owns one leg and its lanes own the 6 rows/components of
// a spatial quantity (lanes 6,7 of a group carry zeros).  The sum lands in every lane of the group.
#define NMF_DPP(v, ctrl) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (v)), (ctrl), 0xf, 0xf, true))
// NMF_PIN: the value is final here — keeps the compiler from contracting the producing multiply into the first DPP step
// (v_mul + v_mov_dpp + v_fmac instead of v_mul + v_add_dpp) or re-associating the last step with what follows.
#define NMF_PIN(v) asm("" : "+v"(v))
__device__ __forceinline__ float grp8_sum(float v) {
  NMF_PIN(v);
  v += NMF_DPP(v, 0xB1);    // quad_perm [1,0,3,2]
  v += NMF_DPP(v, 0x4E);    // quad_perm [2,3,0,1]
  v += NMF_DPP(v, 0x141);   // row_half_mirror: lane i <-> 7-i inside each 8-lane half row
  NMF_PIN(v);
  return v;
}

// ---------------------------------------------------------------- wave reductions
// 64-lane sum on the VALU (DPP), result broadcast through an SGPR: no LDS-crossbar round trips.
#define NMF_DPP_ROWS(v, ctrl, rows) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (v)), (ctrl), (rows), 0xf, false))
__device__ __forceinline__ float wave_sum(float v) {
  v += NMF_DPP(v, 0xB1);            // xor 1
  v += NMF_DPP(v, 0x4E);            // xor 2
  v += NMF_DPP(v, 0x141);           // 8-lane halves
  v += NMF_DPP(v, 0x140);           // row_mirror: every lane of a 16-lane row holds the row sum
  v += NMF_DPP_ROWS(v, 0x142, 0xa); // row_bcast15 into rows 1 and 3
  v += NMF_DPP_ROWS(v, 0x143, 0xc); // row_bcast31 into rows 2 and 3
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
// broadcast lane `i` (0..7) of every 8-lane group to the whole group (LDS crossbar, no memory)
template <int I>
__device__ __forceinline__ float grp8_bcast(float v) {
  // ds_swizzle bit-mask mode: lane' = ((lane & and) | or) ^ xor over 32-lane halves
  constexpr int pattern = 0x18 | (I << 5);
  return __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), pattern));
}
// wave-wide min / max on the VALU (same DPP ladder as wave_sum); `idn` is the identity fed to the
// rows a row_bcast step does not write
#define NMF_DPP_OLD(old, v, ctrl, rows) __builtin_amdgcn_update_dpp((old), (v), (ctrl), (rows), 0xf, false)
template <class Op>
__device__ __forceinline__ int wave_reduce_bits(int v, int idn, Op op) {
  v = op(v, NMF_DPP_OLD(idn, v, 0xB1, 0xf));
  v = op(v, NMF_DPP_OLD(idn, v, 0x4E, 0xf));
  v = op(v, NMF_DPP_OLD(idn, v, 0x141, 0xf));
  v = op(v, NMF_DPP_OLD(idn, v, 0x140, 0xf));
  v = op(v, NMF_DPP_OLD(idn, v, 0x142, 0xa));
  v = op(v, NMF_DPP_OLD(idn, v, 0x143, 0xc));
  return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ float wave_min(float v) {
  return __builtin_bit_cast(float, wave_reduce_bits(__builtin_bit_cast(int, v), 0x7f800000, [](int a, int b) {
    return __builtin_bit_cast(int, fminf(__builtin_bit_cast(float, a), __builtin_bit_cast(float, b))); }));
}
__device__ __forceinline__ float wave_max(float v) {
  return __builtin_bit_cast(float, wave_reduce_bits(__builtin_bit_cast(int, v), (int)0xff800000, [](int a, int b) {
    return __builtin_bit_cast(int, fmaxf(__builtin_bit_cast(float, a), __builtin_bit_cast(float, b))); }));
}
// Sum over the wave's eight 8-lane groups, lane r of every group ending up with the total of lanes r of all groups: three VALU
// steps — rotate by 8 inside the 16-lane rows (DPP), then the gfx950 row and half swaps (v_permlane16_swap / v_permlane32_swap:
// after swap(v, v) the two results hold each lane's own and its partner row's / half's value).
__device__ __forceinline__ float groups_sum(float v) {
  v += NMF_DPP(v, 0x128);                                                           // row_ror:8
  // (inline assembly: with the same value in both operands the compiler's builtin returned the first result twice — ROCm 7.2;
  // scripts/micro/groups_sum_test.hip checks the three steps on the GPU)
  { float a = v, b = v; asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b)); v = a + b; }
  { float a = v, b = v; asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b)); v = a + b; }
  return v;
}
// inclusive prefix minimum over the wave's lanes 0..lane (row_shr ladder inside the 16-lane rows, then the rows' totals)
__device__ __forceinline__ int wave_prefix_min_int(int v) {
  constexpr int big = 0x7fffffff;
  auto mn = [](int a, int b) { return a < b ? a : b; };
  v = mn(v, NMF_DPP_OLD(big, v, 0x111, 0xf));      // row_shr:1 (lanes without a source keep the identity)
  v = mn(v, NMF_DPP_OLD(big, v, 0x112, 0xf));
  v = mn(v, NMF_DPP_OLD(big, v, 0x114, 0xf));
  v = mn(v, NMF_DPP_OLD(big, v, 0x118, 0xf));
  v = mn(v, NMF_DPP_OLD(big, v, 0x142, 0xa));      // row_bcast15 into rows 1 and 3
  v = mn(v, NMF_DPP_OLD(big, v, 0x143, 0xc));      // row_bcast31 into rows 2 and 3
  return v;
}
__device__ __forceinline__ int wave_min_int(int v) {
  return wave_reduce_bits(v, 0x7fffffff, [](int a, int b) { return a < b ? a : b; });
}
// value extremum with the lowest index among ties (two VALU reductions; no ds_bpermute)
__device__ __forceinline__ void wave_argmin(float& v, int& i) {
  const float m = wave_min(v);
  i = wave_min_int(v == m ? i : 0x7fffffff);
  v = m;
}
__device__ __forceinline__ void wave_argmax(float& v, int& i) {
  const float m = wave_max(v);
  i = wave_min_int(v == m ? i : 0x7fffffff);
  v = m;
}

}  // namespace nmf
