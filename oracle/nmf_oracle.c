/*
 * nmf_oracle.c — CPU restatement of the NeuroMechFly physics step.  TEST INFRASTRUCTURE.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library; the product (flygym_amd/) never does.
 *
 * What it restates.  The reference executes one physics step by calling a third-party
 * engine: `mujoco.mj_step` (reference src/flygym/simulation.py:74-76, package
 * mujoco==3.6.0 pinned in uv.lock:1335-1336) or `mujoco_warp.step`
 * (src/flygym/warp/simulation.py:260-263, mujoco-warp==3.6.0).  Neither package is in
 * /root/reference or installable here, so this file restates the *published* algorithm
 * of that pipeline (MuJoCo documentation, "Computation" chapter) for the model the
 * reference builds, with the options the reference selects:
 *   - src/flygym/assets/model/mujoco_globals.yaml:9-19  (Euler, Newton, dt 1e-4, gravity)
 *   - src/flygym/compose/fly.py:221-299 (hinges: stiffness/damping/armature/springref)
 *   - src/flygym/compose/fly.py:301-369, 407-441 (position + adhesion actuators)
 *   - src/flygym/compose/world.py:291-331 (explicit geom–plane pairs, contact sensors)
 *   - src/flygym/compose/physics.py:61-111 (friction/solref/solimp/margin)
 *   - src/flygym/warp/simulation.py:427-448 (noslip stripped on the batched path)
 *
 * PARITY UNPINNED: the reference's tests hold no numeric dynamics vectors and
 * the engine cannot be run here, so agreement with real MuJoCo 3.6.0 is untested.  The
 * oracle is pinned instead by first-principles known-answer tests (tests/test_oracle_*.py):
 * mass matrix against Σ JᵀIJ, bias forces against a numerical Lagrangian, free fall,
 * static weight balance, energy, and the reference's own invariants.
 *
 * Stage order (MuJoCo mj_step = mj_forward + mj_Euler):
 *   kinematics → inertias about a reference point → CRBA → LᵀDL → collision →
 *   constraint rows (pyramidal, condim 3) → adhesion transmission → velocities →
 *   passive → RNE bias → actuation → smooth acceleration → Newton constraint solve →
 *   contact sensors → implicit-damping Euler.
 *
 * Build: -DNMF_REAL=double (default) or float; symbols are suffixed _f64 / _f32.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#ifndef NMF_REAL_IS_FLOAT
typedef double real;
#define SFX(n) n##_f64
#define R_SQRT sqrt
#define R_FABS fabs
#define R_SIN sin
#define R_COS cos
#define R_POW pow
#define R_EXP exp
#define R_EPS 2.220446049250313e-16
#else
typedef float real;
#define SFX(n) n##_f32
#define R_SQRT sqrtf
#define R_FABS fabsf
#define R_SIN sinf
#define R_COS cosf
#define R_EXP expf
#define R_POW powf
#define R_EPS 1.1920929e-07f
#endif

#define NMF_MAXCON 48
#ifndef NMF_NOISE_FACTOR
#define NMF_NOISE_FACTOR 8
#endif
#define NMF_MINVAL 1e-15
#define NMF_MAXEFC (4 * NMF_MAXCON + 6)   /* contact rows + the 6 rows of the tether weld */
#define GEOM_CAPSULE 0
#define GEOM_HULL 1
#define ACT_POSITION 0
#define ACT_ADHESION 1
#define ACT_MOTOR 2

/* ------------------------------------------------------------------ blob */
typedef struct {
  char name[32];
  uint32_t dtype, ndim;
  int64_t shape[4];
  int64_t offset, nbytes;
} blob_entry;

static const blob_entry* blob_find(const uint8_t* blob, const char* name) {
  uint32_t n;
  memcpy(&n, blob + 12, 4);
  const blob_entry* e = (const blob_entry*)(blob + 16);
  for (uint32_t i = 0; i < n; i++)
    if (strncmp(e[i].name, name, 32) == 0) return &e[i];
  return NULL;
}

static int64_t entry_count(const blob_entry* e) {
  int64_t c = 1;
  for (uint32_t i = 0; i < e->ndim; i++) c *= e->shape[i];
  return c;
}

static real* blob_real(const uint8_t* blob, const char* name, int64_t* count) {
  const blob_entry* e = blob_find(blob, name);
  if (!e || e->dtype != 0) { fprintf(stderr, "nmf_oracle: missing f64 entry %s\n", name); abort(); }
  int64_t c = entry_count(e);
  real* out = (real*)malloc(sizeof(real) * (size_t)(c > 0 ? c : 1));
  const double* src = (const double*)(blob + e->offset);
  for (int64_t i = 0; i < c; i++) out[i] = (real)src[i];
  if (count) *count = c;
  return out;
}

static int* blob_int(const uint8_t* blob, const char* name, int64_t* count) {
  const blob_entry* e = blob_find(blob, name);
  if (!e || e->dtype != 1) { fprintf(stderr, "nmf_oracle: missing i32 entry %s\n", name); abort(); }
  int64_t c = entry_count(e);
  int* out = (int*)malloc(sizeof(int) * (size_t)(c > 0 ? c : 1));
  memcpy(out, blob + e->offset, sizeof(int) * (size_t)c);
  if (count) *count = c;
  return out;
}

/* ------------------------------------------------------------------ model */
typedef struct {
  int nb, nv, nq, nu, ng, nseg, nsite, nhv, nsensor;
  real timestep, gravity[3], tolerance, hull_skin, plane[4], meaninertia, terrain[5];
  int terrain_type;
  /* tether: soft weld of the root body to its spawn pose (reference compose/world.py:358-365) */
  int weld_active; real weld_pos[3], weld_quat[4], weld_solref[2], weld_solimp[5], weld_invweight[2];
  int max_iter;
  int noslip_iter;                     /* option/noslip_iterations (mujoco_globals.yaml:15); 0 on the batched path */
  int *body_parent, *body_dofadr, *body_dofnum;
  real *body_pos, *body_quat, *body_mass, *body_ipos, *body_inertia;
  int *dof_body, *dof_parent;
  real *dof_axis, *dof_armature, *dof_damping, *dof_stiffness, *dof_springref;
  int* seg_body; real *seg_pos, *seg_quat;
  int* site_body; real* site_pos;
  int *act_type, *act_trn, *act_limited, *act_geom;
  real *act_gain, *act_bias, *act_forcerange, *act_ctrlrange;
  /* optional blob entry act_general [nu][NMF_ACTGEN]: MuJoCo's general actuator for the types beyond the affine stateless ones
   * (intvelocity, damper, cylinder, muscle; reference compose/fly.py:65-77, 301-369) — NULL when the model has none.  Row layout
   * (flygym_amd/compiler/model.py::_general_row): 0 flags (1 on | 2 forcelimited | dof << 8; the legacy act_limited[u][0] is 0 for these), 1 dyntype (0 none, 1 integrator, 2 filter, 3 filterexact, 4 muscle),
   * 2 gaintype (0 fixed, 1 affine, 2 muscle), 3 biastype (0 none, 1 affine, 2 muscle), 4 actlimited, 5 gear, 6..8 dynprm,
   * 9..17 gainprm, 18..26 biasprm, 27..28 actrange, 29..30 lengthrange, 31 acc0 */
  real *act_general;
  real *key_qpos, *key_ctrl;
  int *geom_body, *geom_type, *geom_hulladr, *geom_hullnum, *geom_sensor;
  real *geom_p0, *geom_p1, *geom_radius, *geom_bsphere, *geom_invweight0, *hull_vert;
  real *pair_friction, *pair_solref, *pair_solimp, *pair_margin;
  /* named engine semantics (blob entry sem_options, flygym_amd.compiler.model.EngineSemantics): the low-confidence
   * rows of SURVEY.md Appendix A as switches, read here and by the HIP kernel alike */
  int sem_pyramid_plain, sem_adhesion_fused, sem_sensor_contact_frame, sem_max_hull_contacts;
  int sem_terrain_walls;    /* terrains: side faces of the cells collide (flygym_amd/compose/world.py::terrain_probe) */
} omodel;

typedef struct {
  /* state */
  real *qpos, *qvel, *ctrl, *qacc_warmstart;
  real *act, *act_next;                /* nu: activation state of the stateful actuators (one slot per actuator; 0 for the stateless) */
  real time;
  /* position-dependent */
  real *xpos, *xquat, *xmat;          /* dynamic bodies */
  real *S;                             /* nv x 6 motion subspace about o (w;v) */
  real *daxis, *danchor;               /* nv x 3 world axis / anchor (hinges) */
  real *Ib;                            /* nb x 10: m, h(3), I_o sym6 */
  real *Ic;                            /* composite */
  real *M, *L;                         /* nv x nv dense storage (tree-sparse content) */
  real *Ld;                            /* nv: D of LᵀDL */
  /* contacts */
  int ncon, nefc, overflow;
  int con_geom[NMF_MAXCON];
  real con_dist[NMF_MAXCON], con_pos[NMF_MAXCON][3], con_frame[NMF_MAXCON][9];
  real con_mu[NMF_MAXCON];
  real *J;                             /* (4*MAXCON) x nv */
  real *Jc;                            /* (3*MAXCON) x nv: normal, t1, t2 Jacobians */
  real efc_D[NMF_MAXEFC], efc_aref[NMF_MAXEFC], efc_force[NMF_MAXEFC];
  real efc_R[NMF_MAXEFC], efc_KBI[NMF_MAXEFC][3], efc_pos[NMF_MAXEFC];
  int efc_bilateral[NMF_MAXEFC];
  /* velocity / force */
  real *cvel, *cacc, *cfrc;            /* nb x 6 */
  real *qfrc_passive, *qfrc_bias, *qfrc_actuator, *qfrc_smooth, *qfrc_constraint;
  real *qacc_smooth, *qacc, *actuator_force, *act_moment; /* act_moment: nu_adh x nv lazily nu x nv */
  real *sensordata;
  /* outputs for the named surface */
  real *seg_xpos, *seg_xquat, *site_xpos;
  /* solver stats */
  int solver_iter;
  real solver_cost;
  /* 0: the stopping rules the HIP kernel shares (tolerance tests + float rounding-floor exits, line search stops when
   *    a 1-D Newton step keeps the active set);
   * 1: MuJoCo-documented rules only — gradient / improvement against `tolerance`, line search iterated to its fixed
   *    point — kept untouched by kernel work so that the converged solution has an independent anchor */
  int noslip_on;                       /* 1: run the model's noslip_iter sweeps (the CPU class); 0: the batched class strips them */
  int max_contacts;                    /* contacts kept per step (<= NMF_MAXCON; HIPSimulation's max_contacts): later ones, in geom order, are dropped and the step flagged */
  int solver_mode;
  /* scratch */
  real *w1, *w2, *w3, *w4, *w5, *H;
} odata;

#define EXPORT __attribute__((visibility("default")))

EXPORT void* SFX(nmfo_model_create)(const uint8_t* blob, int64_t nbytes) {
  (void)nbytes;
  if (memcmp(blob, "NMFMODEL", 8) != 0) return NULL;
  { uint32_t version; memcpy(&version, blob + 8, 4); if (version != 4) return NULL; }   /* older blobs lack entries read below */
  omodel* m = (omodel*)calloc(1, sizeof(omodel));
  int64_t c;
  m->body_parent = blob_int(blob, "body_parent", &c); m->nb = (int)c;
  m->body_dofadr = blob_int(blob, "body_dofadr", NULL);
  m->body_dofnum = blob_int(blob, "body_dofnum", NULL);
  m->body_pos = blob_real(blob, "body_pos", NULL);
  m->body_quat = blob_real(blob, "body_quat", NULL);
  m->body_mass = blob_real(blob, "body_mass", NULL);
  m->body_ipos = blob_real(blob, "body_ipos", NULL);
  m->body_inertia = blob_real(blob, "body_inertia", NULL);
  m->dof_body = blob_int(blob, "dof_body", &c); m->nv = (int)c; m->nq = m->nv + 1;
  m->dof_parent = blob_int(blob, "dof_parent", NULL);
  m->dof_axis = blob_real(blob, "dof_axis", NULL);
  m->dof_armature = blob_real(blob, "dof_armature", NULL);
  m->dof_damping = blob_real(blob, "dof_damping", NULL);
  m->dof_stiffness = blob_real(blob, "dof_stiffness", NULL);
  m->dof_springref = blob_real(blob, "dof_springref", NULL);
  m->seg_body = blob_int(blob, "seg_body", &c); m->nseg = (int)c;
  m->seg_pos = blob_real(blob, "seg_pos", NULL);
  m->seg_quat = blob_real(blob, "seg_quat", NULL);
  m->site_body = blob_int(blob, "site_body", &c); m->nsite = (int)c;
  m->site_pos = blob_real(blob, "site_pos", NULL);
  m->act_type = blob_int(blob, "act_type", &c); m->nu = (int)c;
  m->act_trn = blob_int(blob, "act_trn", NULL);
  m->act_limited = blob_int(blob, "act_limited", NULL);
  m->act_geom = blob_int(blob, "act_geom", NULL);
  { int* so = blob_int(blob, "sem_options", NULL);
    m->sem_pyramid_plain = so[0]; m->sem_adhesion_fused = so[1]; m->sem_sensor_contact_frame = so[2];
    m->sem_max_hull_contacts = so[3] >= 1 && so[3] <= 4 ? so[3] : 4; m->sem_terrain_walls = so[4]; free(so); }
  m->act_gain = blob_real(blob, "act_gain", NULL);
  m->act_bias = blob_real(blob, "act_bias", NULL);
  m->act_forcerange = blob_real(blob, "act_forcerange", NULL);
  m->act_ctrlrange = blob_real(blob, "act_ctrlrange", NULL);
  m->act_general = blob_find(blob, "act_general") ? blob_real(blob, "act_general", NULL) : NULL;
  m->key_qpos = blob_real(blob, "key_qpos", NULL);
  m->key_ctrl = blob_real(blob, "key_ctrl", NULL);
  m->geom_body = blob_int(blob, "geom_body", &c); m->ng = (int)c;
  m->geom_type = blob_int(blob, "geom_type", NULL);
  m->geom_hulladr = blob_int(blob, "geom_hulladr", NULL);
  m->geom_hullnum = blob_int(blob, "geom_hullnum", NULL);
  m->geom_sensor = blob_int(blob, "geom_sensor", NULL);
  m->geom_p0 = blob_real(blob, "geom_p0", NULL);
  m->geom_p1 = blob_real(blob, "geom_p1", NULL);
  m->geom_radius = blob_real(blob, "geom_radius", NULL);
  m->geom_bsphere = blob_real(blob, "geom_bsphere", NULL);
  m->geom_invweight0 = blob_real(blob, "geom_invweight0", NULL);
  m->hull_vert = blob_real(blob, "hull_vert", &c); m->nhv = (int)(c / 3);
  m->pair_friction = blob_real(blob, "pair_friction", NULL);
  m->pair_solref = blob_real(blob, "pair_solref", NULL);
  m->pair_solimp = blob_real(blob, "pair_solimp", NULL);
  m->pair_margin = blob_real(blob, "pair_margin", NULL);
  real* t;
  t = blob_real(blob, "opt_timestep", NULL); m->timestep = t[0]; free(t);
  t = blob_real(blob, "opt_gravity", NULL); memcpy(m->gravity, t, 3 * sizeof(real)); free(t);
  t = blob_real(blob, "opt_tolerance", NULL); m->tolerance = t[0]; free(t);
  t = blob_real(blob, "hull_skin", NULL); m->hull_skin = t[0]; free(t);
  t = blob_real(blob, "plane", NULL); memcpy(m->plane, t, 4 * sizeof(real)); free(t);
  t = blob_real(blob, "terrain_params", NULL); memcpy(m->terrain, t, 5 * sizeof(real)); free(t);
  { int* tt = blob_int(blob, "terrain_type", NULL); m->terrain_type = tt[0]; free(tt); }
  { int* wa = blob_int(blob, "weld_active", NULL); m->weld_active = wa[0]; free(wa);
    t = blob_real(blob, "weld_params", NULL);    /* pos3 quat4 solref2 solimp5 invweight2 */
    memcpy(m->weld_pos, t, 3 * sizeof(real)); memcpy(m->weld_quat, t + 3, 4 * sizeof(real));
    memcpy(m->weld_solref, t + 7, 2 * sizeof(real)); memcpy(m->weld_solimp, t + 9, 5 * sizeof(real));
    memcpy(m->weld_invweight, t + 14, 2 * sizeof(real)); free(t); }
  t = blob_real(blob, "stat_meaninertia", NULL); m->meaninertia = t[0]; free(t);
  int64_t n_opt = 0;
  int* it = blob_int(blob, "opt_solver", &n_opt); m->max_iter = it[0]; m->noslip_iter = n_opt > 1 ? it[1] : 0; free(it);
  it = blob_int(blob, "n_sensor", NULL); m->nsensor = it[0]; free(it);
  return m;
}

EXPORT void SFX(nmfo_model_dims)(const void* mv, int* out) {
  const omodel* m = (const omodel*)mv;
  out[0] = m->nq; out[1] = m->nv; out[2] = m->nu; out[3] = m->nb; out[4] = m->nseg;
  out[5] = m->ng; out[6] = m->nsite; out[7] = NMF_MAXCON; out[8] = m->nsensor; out[9] = (int)sizeof(real);
}

#define ALLOC(n) ((real*)calloc((size_t)((n) > 0 ? (n) : 1), sizeof(real)))

EXPORT void* SFX(nmfo_data_create)(const void* mv) {
  const omodel* m = (const omodel*)mv;
  odata* d = (odata*)calloc(1, sizeof(odata));
  d->max_contacts = NMF_MAXCON;
  int nv = m->nv, nb = m->nb;
  d->qpos = ALLOC(m->nq); d->qvel = ALLOC(nv); d->ctrl = ALLOC(m->nu); d->qacc_warmstart = ALLOC(nv);
  d->act = ALLOC(m->nu); d->act_next = ALLOC(m->nu);
  d->xpos = ALLOC(nb * 3); d->xquat = ALLOC(nb * 4); d->xmat = ALLOC(nb * 9);
  d->S = ALLOC(nv * 6); d->daxis = ALLOC(nv * 3); d->danchor = ALLOC(nv * 3);
  d->Ib = ALLOC(nb * 10); d->Ic = ALLOC(nb * 10);
  d->M = ALLOC(nv * nv); d->L = ALLOC(nv * nv); d->Ld = ALLOC(nv); d->H = ALLOC(nv * nv);
  d->J = ALLOC(NMF_MAXEFC * nv); d->Jc = ALLOC(3 * NMF_MAXCON * nv);
  d->cvel = ALLOC(nb * 6); d->cacc = ALLOC(nb * 6); d->cfrc = ALLOC(nb * 6);
  d->qfrc_passive = ALLOC(nv); d->qfrc_bias = ALLOC(nv); d->qfrc_actuator = ALLOC(nv);
  d->qfrc_smooth = ALLOC(nv); d->qfrc_constraint = ALLOC(nv); d->qacc_smooth = ALLOC(nv);
  d->qacc = ALLOC(nv); d->actuator_force = ALLOC(m->nu); d->act_moment = ALLOC(m->nu * nv);
  d->sensordata = ALLOC(16 * 6);
  d->seg_xpos = ALLOC(m->nseg * 3); d->seg_xquat = ALLOC(m->nseg * 4); d->site_xpos = ALLOC(m->nsite * 3);
  d->w1 = ALLOC(nv); d->w2 = ALLOC(nv); d->w3 = ALLOC(nv); d->w4 = ALLOC(nv); d->w5 = ALLOC(NMF_MAXEFC + nv);
  return d;
}

/* ------------------------------------------------------------------ small math */
static inline void cross3(real* r, const real* a, const real* b) {
  real x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
  r[0] = x; r[1] = y; r[2] = z;
}
static inline real dot3(const real* a, const real* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static inline void quat_mul(real* r, const real* a, const real* b) {
  real w = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
  real x = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
  real y = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
  real z = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
  r[0] = w; r[1] = x; r[2] = y; r[3] = z;
}
static inline void quat_norm(real* q) {
  real n = R_SQRT(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  if (n < (real)NMF_MINVAL) { q[0] = 1; q[1] = q[2] = q[3] = 0; return; }
  real s = (real)1 / n; q[0] *= s; q[1] *= s; q[2] *= s; q[3] *= s;
}
static inline void quat_to_mat(real* m, const real* q) {
  real w = q[0], x = q[1], y = q[2], z = q[3];
  m[0] = 1 - 2 * (y * y + z * z); m[1] = 2 * (x * y - w * z); m[2] = 2 * (x * z + w * y);
  m[3] = 2 * (x * y + w * z); m[4] = 1 - 2 * (x * x + z * z); m[5] = 2 * (y * z - w * x);
  m[6] = 2 * (x * z - w * y); m[7] = 2 * (y * z + w * x); m[8] = 1 - 2 * (x * x + y * y);
}
static inline void mat_vec(real* r, const real* m, const real* v) {
  real x = m[0] * v[0] + m[1] * v[1] + m[2] * v[2];
  real y = m[3] * v[0] + m[4] * v[1] + m[5] * v[2];
  real z = m[6] * v[0] + m[7] * v[1] + m[8] * v[2];
  r[0] = x; r[1] = y; r[2] = z;
}
static inline void matT_vec(real* r, const real* m, const real* v) {
  real x = m[0] * v[0] + m[3] * v[1] + m[6] * v[2];
  real y = m[1] * v[0] + m[4] * v[1] + m[7] * v[2];
  real z = m[2] * v[0] + m[5] * v[1] + m[8] * v[2];
  r[0] = x; r[1] = y; r[2] = z;
}
static inline void axis_angle_quat(real* q, const real* axis, real angle) {
  real h = (real)0.5 * angle, s = R_SIN(h);
  q[0] = R_COS(h); q[1] = axis[0] * s; q[2] = axis[1] * s; q[3] = axis[2] * s;
}
/* spatial inertia (m, h, I sym6 = xx yy zz xy xz yz) times motion (w;v) → force (n;f) */
static inline void inert_mul(real* F, const real* I, const real* V) {
  const real m = I[0], *h = I + 1, *J = I + 4, *w = V, *v = V + 3;
  real hv[3], hw[3];
  cross3(hv, h, v); cross3(hw, h, w);
  F[0] = J[0] * w[0] + J[3] * w[1] + J[4] * w[2] + hv[0];
  F[1] = J[3] * w[0] + J[1] * w[1] + J[5] * w[2] + hv[1];
  F[2] = J[4] * w[0] + J[5] * w[1] + J[2] * w[2] + hv[2];
  F[3] = m * v[0] - hw[0]; F[4] = m * v[1] - hw[1]; F[5] = m * v[2] - hw[2];
}
static inline void cross_motion(real* r, const real* a, const real* b) { /* a x b, motion vectors */
  real t0[3], t1[3], t2[3];
  cross3(t0, a, b); cross3(t1, a, b + 3); cross3(t2, a + 3, b);
  r[0] = t0[0]; r[1] = t0[1]; r[2] = t0[2];
  r[3] = t1[0] + t2[0]; r[4] = t1[1] + t2[1]; r[5] = t1[2] + t2[2];
}
static inline void cross_force(real* r, const real* v, const real* F) { /* v x* F */
  real t0[3], t1[3], t2[3];
  cross3(t0, v, F); cross3(t1, v + 3, F + 3); cross3(t2, v, F + 3);
  r[0] = t0[0] + t1[0]; r[1] = t0[1] + t1[1]; r[2] = t0[2] + t1[2];
  r[3] = t2[0]; r[4] = t2[1]; r[5] = t2[2];
}
static inline real dot6(const real* a, const real* b) {
  return a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3] + a[4] * b[4] + a[5] * b[5];
}

/* ------------------------------------------------------------------ stage: kinematics */
static void kinematics(const omodel* m, odata* d) {
  for (int b = 0; b < m->nb; b++) {
    real* pos = d->xpos + 3 * b; real* quat = d->xquat + 4 * b;
    int p = m->body_parent[b];
    if (p < 0) {
      pos[0] = d->qpos[0]; pos[1] = d->qpos[1]; pos[2] = d->qpos[2];
      quat[0] = d->qpos[3]; quat[1] = d->qpos[4]; quat[2] = d->qpos[5]; quat[3] = d->qpos[6];
      quat_norm(quat);
      quat_to_mat(d->xmat + 9 * b, quat);
      for (int i = 0; i < 3; i++) {
        real* a = d->daxis + 3 * i; a[0] = a[1] = a[2] = 0; a[i] = 1;
        real* ar = d->daxis + 3 * (3 + i);
        ar[0] = d->xmat[9 * b + i]; ar[1] = d->xmat[9 * b + 3 + i]; ar[2] = d->xmat[9 * b + 6 + i];
      }
      for (int i = 0; i < 6; i++) memcpy(d->danchor + 3 * i, pos, 3 * sizeof(real));
      continue;
    }
    real off[3];
    mat_vec(off, d->xmat + 9 * p, m->body_pos + 3 * b);
    pos[0] = d->xpos[3 * p] + off[0]; pos[1] = d->xpos[3 * p + 1] + off[1]; pos[2] = d->xpos[3 * p + 2] + off[2];
    quat_mul(quat, d->xquat + 4 * p, m->body_quat + 4 * b);
    int adr = m->body_dofadr[b];
    for (int j = adr; j < adr + m->body_dofnum[b]; j++) {
      /* axis is fixed in the frame reached after this body's previous hinges; pos = 0 anchors */
      real rm[9]; quat_to_mat(rm, quat);
      mat_vec(d->daxis + 3 * j, rm, m->dof_axis + 3 * j);
      memcpy(d->danchor + 3 * j, pos, 3 * sizeof(real));
      real qj[4], t[4];
      axis_angle_quat(qj, m->dof_axis + 3 * j, d->qpos[j + 1]);
      quat_mul(t, quat, qj);
      memcpy(quat, t, 4 * sizeof(real));
    }
    quat_norm(quat);
    quat_to_mat(d->xmat + 9 * b, quat);
  }
  /* motion subspaces about o = root origin */
  const real* o = d->xpos;
  for (int j = 0; j < m->nv; j++) {
    real* S = d->S + 6 * j; const real* a = d->daxis + 3 * j;
    if (j < 3) { S[0] = S[1] = S[2] = 0; S[3] = a[0]; S[4] = a[1]; S[5] = a[2]; continue; }
    real r[3] = {o[0] - d->danchor[3 * j], o[1] - d->danchor[3 * j + 1], o[2] - d->danchor[3 * j + 2]};
    S[0] = a[0]; S[1] = a[1]; S[2] = a[2];
    cross3(S + 3, a, r);
  }
}

static void named_poses(const omodel* m, odata* d) {
  for (int s = 0; s < m->nseg; s++) {
    int b = m->seg_body[s]; real off[3];
    mat_vec(off, d->xmat + 9 * b, m->seg_pos + 3 * s);
    for (int k = 0; k < 3; k++) d->seg_xpos[3 * s + k] = d->xpos[3 * b + k] + off[k];
    quat_mul(d->seg_xquat + 4 * s, d->xquat + 4 * b, m->seg_quat + 4 * s);
    quat_norm(d->seg_xquat + 4 * s);
  }
  for (int s = 0; s < m->nsite; s++) {
    int b = m->site_body[s]; real off[3];
    mat_vec(off, d->xmat + 9 * b, m->site_pos + 3 * s);
    for (int k = 0; k < 3; k++) d->site_xpos[3 * s + k] = d->xpos[3 * b + k] + off[k];
  }
}

/* ------------------------------------------------------------------ stage: inertia, CRBA, factor */
static void body_inertias(const omodel* m, odata* d) {
  const real* o = d->xpos;
  for (int b = 0; b < m->nb; b++) {
    const real* R = d->xmat + 9 * b; const real* s = m->body_inertia + 6 * b;
    real Il[9] = {s[0], s[3], s[4], s[3], s[1], s[5], s[4], s[5], s[2]};
    real T[9], Iw[9];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
      real a = 0; for (int k = 0; k < 3; k++) a += R[3 * i + k] * Il[3 * k + j]; T[3 * i + j] = a; }
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
      real a = 0; for (int k = 0; k < 3; k++) a += T[3 * i + k] * R[3 * j + k]; Iw[3 * i + j] = a; }
    real c[3]; mat_vec(c, R, m->body_ipos + 3 * b);
    for (int k = 0; k < 3; k++) c[k] += d->xpos[3 * b + k] - o[k];
    real ms = m->body_mass[b], cc = dot3(c, c);
    real* I = d->Ib + 10 * b;
    I[0] = ms; I[1] = ms * c[0]; I[2] = ms * c[1]; I[3] = ms * c[2];
    I[4] = Iw[0] + ms * (cc - c[0] * c[0]); I[5] = Iw[4] + ms * (cc - c[1] * c[1]); I[6] = Iw[8] + ms * (cc - c[2] * c[2]);
    I[7] = Iw[1] - ms * c[0] * c[1]; I[8] = Iw[2] - ms * c[0] * c[2]; I[9] = Iw[5] - ms * c[1] * c[2];
  }
}

static void crba(const omodel* m, odata* d) {
  int nv = m->nv;
  memcpy(d->Ic, d->Ib, sizeof(real) * 10 * (size_t)m->nb);
  for (int b = m->nb - 1; b > 0; b--) {
    int p = m->body_parent[b];
    for (int k = 0; k < 10; k++) d->Ic[10 * p + k] += d->Ic[10 * b + k];
  }
  memset(d->M, 0, sizeof(real) * (size_t)nv * nv);
  for (int i = 0; i < nv; i++) {
    real F[6]; inert_mul(F, d->Ic + 10 * m->dof_body[i], d->S + 6 * i);
    for (int j = i; j >= 0; j = m->dof_parent[j]) {
      real v = dot6(d->S + 6 * j, F);
      d->M[i * nv + j] = v; d->M[j * nv + i] = v;
    }
    d->M[i * nv + i] += m->dof_armature[i];
  }
}

/* tree-sparse LᵀDL (Featherstone): A (symmetric, ancestor sparsity) → unit-lower L in rows, D */
static void factor_tree(const omodel* m, const real* A, real* L, real* D) {
  int nv = m->nv;
  memcpy(L, A, sizeof(real) * (size_t)nv * nv);
  for (int k = nv - 1; k >= 0; k--) {
    real dk = L[k * nv + k];
    D[k] = dk;
    real inv = (real)1 / dk;
    for (int i = m->dof_parent[k]; i >= 0; i = m->dof_parent[i]) {
      real a = L[k * nv + i] * inv;
      for (int j = i; j >= 0; j = m->dof_parent[j]) L[i * nv + j] -= a * L[k * nv + j];
      L[k * nv + i] = a;
    }
  }
}
static void solve_tree(const omodel* m, const real* L, const real* D, real* x) {
  int nv = m->nv;
  for (int i = nv - 1; i >= 0; i--)
    for (int j = m->dof_parent[i]; j >= 0; j = m->dof_parent[j]) x[j] -= L[i * nv + j] * x[i];
  for (int i = 0; i < nv; i++) x[i] /= D[i];
  for (int i = 0; i < nv; i++)
    for (int j = m->dof_parent[i]; j >= 0; j = m->dof_parent[j]) x[i] -= L[i * nv + j] * x[j];
}
static void mul_M(const omodel* m, const real* M, const real* x, real* y) {
  int nv = m->nv;
  for (int i = 0; i < nv; i++) y[i] = M[i * nv + i] * x[i];
  for (int i = 0; i < nv; i++)
    for (int j = m->dof_parent[i]; j >= 0; j = m->dof_parent[j]) {
      y[i] += M[i * nv + j] * x[j]; y[j] += M[i * nv + j] * x[i];
    }
}

/* ------------------------------------------------------------------ stage: collision */
static void make_frame(real* frame, const real* n) {
  /* x = normal; first tangent from the world axis least aligned with it (mju_makeFrame rule) */
  real t[3] = {0, 0, 0};
  if (R_FABS(n[1]) < (real)0.5) t[1] = 1; else t[2] = 1;
  real dn = dot3(t, n);
  real t1[3] = {t[0] - dn * n[0], t[1] - dn * n[1], t[2] - dn * n[2]};
  real l = R_SQRT(dot3(t1, t1));
  t1[0] /= l; t1[1] /= l; t1[2] /= l;
  real t2[3]; cross3(t2, n, t1);
  memcpy(frame, n, 3 * sizeof(real)); memcpy(frame + 3, t1, 3 * sizeof(real)); memcpy(frame + 6, t2, 3 * sizeof(real));
}

static void add_contact(const omodel* m, odata* d, int g, real dist, const real* pos_surface, const real* n) {
  if (d->ncon >= d->max_contacts) { d->overflow = 1; return; }
  int c = d->ncon++;
  d->con_geom[c] = g; d->con_dist[c] = dist;
  for (int k = 0; k < 3; k++) d->con_pos[c][k] = pos_surface[k] - (real)0.5 * dist * n[k];
  make_frame(d->con_frame[c], n);
  d->con_mu[c] = m->pair_friction[5 * g];
}

/* piecewise-constant ground height under (x, y): build-defined terrains (flygym_amd/compose/world.py) */
#ifdef NMF_REAL_IS_FLOAT
#define R_FLOOR floorf
#else
#define R_FLOOR floor
#endif
static real terrain_kind(int kind, const real* p, real x, real y) {
  if (kind == 1) { real period = p[0] + p[1]; real u = x - R_FLOOR(x / period) * period; return u < p[0] ? (real)0 : -p[2]; }
  if (kind == 2) { real i = R_FLOOR(x / p[0]), j = R_FLOOR(y / p[0]); real sum = i + j; real par = sum - 2 * R_FLOOR(sum / 2);
                   return par != 0 ? p[1] : (real)0; }
  return 0;
}
static real terrain_height(const omodel* m, real x, real y) {
  const real* p = m->terrain;
  if (m->terrain_type == 3) {
    real st = R_FLOOR(x / p[3]); real k = st - 3 * R_FLOOR(st / 3);
    real gp[3] = {(real)1.0, p[1], p[2]}, bp[2] = {p[0], (real)0.35};
    return k == 1 ? terrain_kind(1, gp, x, y) : (k == 2 ? terrain_kind(2, bp, x, y) : (real)0);
  }
  return terrain_kind(m->terrain_type, p, x, y);
}


/* The terrain as boxes: bounds (x_lo, x_hi, y_lo, y_hi) of the constant-height cell that holds (x, y); +-NMF_FAR where the
 * lattice does not divide that axis.  Restates flygym_amd/compose/world.py::_cell_bounds. */
#define NMF_FAR ((real)1e30)
static void cell_bounds_kind(int kind, const real* p, real x, real y, real* b) {
  b[0] = -NMF_FAR; b[1] = NMF_FAR; b[2] = -NMF_FAR; b[3] = NMF_FAR;
  if (kind == 1) {
    real period = p[0] + p[1]; real k = R_FLOOR(x / period); real u = x - k * period;
    if (u < p[0]) { b[0] = k * period; b[1] = k * period + p[0]; } else { b[0] = k * period + p[0]; b[1] = (k + 1) * period; }
  } else if (kind == 2) {
    real i = R_FLOOR(x / p[0]), j = R_FLOOR(y / p[0]);
    b[0] = i * p[0]; b[1] = (i + 1) * p[0]; b[2] = j * p[0]; b[3] = (j + 1) * p[0];
  }
}
static void cell_bounds(const omodel* m, real x, real y, real* b) {
  const real* p = m->terrain;
  if (m->terrain_type == 3) {
    real st = R_FLOOR(x / p[3]); real k = st - 3 * R_FLOOR(st / 3);
    real gp[3] = {(real)1.0, p[1], p[2]}, bp[2] = {p[0], (real)0.35};
    cell_bounds_kind(k == 1 ? 1 : (k == 2 ? 2 : 0), k == 1 ? gp : bp, x, y, b);
    real lo = st * p[3], hi = (st + 1) * p[3];
    if (b[0] < lo) b[0] = lo;
    if (b[1] > hi) b[1] = hi;
    return;
  }
  cell_bounds_kind(m->terrain_type, p, x, y, b);
}
/* One collision probe (point, rho = 0, or sphere of radius rho at pw; z measured from the ground plane) against the
 * terrain's boxes: *dtop = signed distance of its lowest point to the top of the cell it is over (NMF_FAR if it is inside
 * that box and leaves it sideways), *dwall / *wall = signed distance to the nearest side face that concerns it and the
 * face's code 1..4 (outward normal +x, -x, +y, -y; NMF_FAR / 0: none).  Restates
 * flygym_amd/compose/world.py::terrain_probe line by line. */
static const real kWallNormal[4][3] = {{1, 0, 0}, {-1, 0, 0}, {0, 1, 0}, {0, -1, 0}};
#define NMF_PROBE_EPS ((real)1e-4)
static void terrain_probe(const omodel* m, const real* pw, real zc, real rho, real* dtop, real* dwall, int* wall) {
  real x = pw[0], y = pw[1];
  real h0 = terrain_height(m, x, y);
  real zb = zc - rho;
  *dtop = zb - h0; *dwall = NMF_FAR; *wall = 0;
  if (!m->sem_terrain_walls) return;
  real b[4]; cell_bounds(m, x, y, b);
  real delta[4] = {b[1] - x, x - b[0], b[3] - y, y - b[2]};
  real he[4];
  he[0] = b[1] < NMF_FAR ? terrain_height(m, b[1] + NMF_PROBE_EPS, y) : h0;
  he[1] = b[0] > -NMF_FAR ? terrain_height(m, b[0] - NMF_PROBE_EPS, y) : h0;
  he[2] = b[3] < NMF_FAR ? terrain_height(m, x, b[3] + NMF_PROBE_EPS) : h0;
  he[3] = b[2] > -NMF_FAR ? terrain_height(m, x, b[2] - NMF_PROBE_EPS) : h0;
  static const int facing[4] = {2, 1, 4, 3}, leaving[4] = {1, 2, 3, 4};
  real edge_top = NMF_FAR;
  for (int e = 0; e < 4; e++) {        /* neighbours that reach above the probe's lowest point */
    if (!(delta[e] < NMF_FAR && he[e] > zb)) continue;
    if (zc - he[e] > delta[e]) { if (zb - he[e] < edge_top) edge_top = zb - he[e]; }   /* over its top edge, nearer the top: that top carries the sphere */
    else if (delta[e] - rho < *dwall) { *dwall = delta[e] - rho; *wall = facing[e]; }  /* its side face */
  }
  if (zb < h0) {
    real pen = h0 - zb; int code = 0;  /* inside its own cell's box: the ways out */
    for (int e = 0; e < 4; e++)
      if (delta[e] < NMF_FAR && he[e] <= zb && delta[e] + rho < pen) { pen = delta[e] + rho; code = leaving[e]; }
    if (code) { *dtop = NMF_FAR; if (-pen < *dwall) { *dwall = -pen; *wall = code; } }
  }
  if (edge_top < *dtop) *dtop = edge_top;
}

/* hull vertex v (body frame) against the terrain: its distance to the top of its cell (NMF_FAR when a side face owns it) */
static real hull_vertex_probe(const omodel* m, const real* R, const real* xp, const real* nb, real c0, const real* v,
                              real* dwall, int* wall) {
  real pw[3]; mat_vec(pw, R, v);
  for (int k = 0; k < 3; k++) pw[k] += xp[k];
  real dtop, dw; int w;
  terrain_probe(m, pw, dot3(nb, v) + c0, (real)0, &dtop, &dw, &w);
  if (dwall) { *dwall = dw; *wall = w; }
  return dtop;
}

static void collide(const omodel* m, odata* d) {
  d->ncon = 0; d->overflow = 0;
  const real* n = m->plane; real pd = m->plane[3];
  for (int g = 0; g < m->ng; g++) {
    int b = m->geom_body[g];
    const real* R = d->xmat + 9 * b; const real* xp = d->xpos + 3 * b;
    real margin = m->pair_margin[g];
    /* bounding sphere cull */
    real cw[3]; mat_vec(cw, R, m->geom_bsphere + 4 * g);
    real dc = dot3(n, cw) + dot3(n, xp) - pd;
    if (dc - m->geom_bsphere[4 * g + 3] - m->terrain[4] > margin) continue;
    if (m->geom_type[g] == GEOM_CAPSULE) {
      /* order within a geom: the contacts with the tops of cells (ground frame) first, then those with side faces — a
       * leg's contact sensor reports the frame of its FIRST contact, i.e. the ground's unless the leg touches walls only */
      real pws[2][3], dwalls[2]; int wls[2];
      for (int e = 0; e < 2; e++) {
        const real* pl = (e == 0 ? m->geom_p0 : m->geom_p1) + 3 * g;
        real* pw = pws[e]; mat_vec(pw, R, pl);
        for (int k = 0; k < 3; k++) pw[k] += xp[k];
        real r = m->geom_radius[g];
        real dist = dot3(n, pw) - pd - r;
        dwalls[e] = NMF_FAR; wls[e] = 0;
        if (m->terrain_type) terrain_probe(m, pw, dot3(n, pw) - pd, r, &dist, &dwalls[e], &wls[e]);
        if (dist <= margin) {
          real ps[3] = {pw[0] - n[0] * r, pw[1] - n[1] * r, pw[2] - n[2] * r};
          add_contact(m, d, g, dist, ps, n);
        }
      }
      for (int e = 0; e < 2; e++) {
        if (wls[e] && dwalls[e] <= margin) {      /* a side face of the terrain: horizontal normal */
          const real* nw = kWallNormal[wls[e] - 1]; const real* pw = pws[e]; real r = m->geom_radius[g];
          real ps[3] = {pw[0] - nw[0] * r, pw[1] - nw[1] * r, pw[2] - nw[2] * r};
          add_contact(m, d, g, dwalls[e], ps, nw);
        }
      }
    } else {
      real nb[3]; matT_vec(nb, R, n);
      real c0 = dot3(n, xp) - pd;
      const real* V = m->hull_vert + 3 * m->geom_hulladr[g];
      int nvv = m->geom_hullnum[g];
      /* vertex distance to the ground under it (flat ground: the plane distance) */
#define VDIST(i) (m->terrain_type ? hull_vertex_probe(m, R, xp, nb, c0, V + 3 * (i), NULL, NULL) : dot3(nb, V + 3 * (i)) + c0)
      int ia = -1; real dmin = 0;
      int iw = -1, wall = 0; real dwmin = NMF_FAR;       /* the vertex nearest to (deepest in) a side face of the terrain */
      for (int i = 0; i < nvv; i++) {
        real di;
        if (m->terrain_type) {
          real dw; int w;
          di = hull_vertex_probe(m, R, xp, nb, c0, V + 3 * i, &dw, &w);
          if (w && dw < dwmin) { dwmin = dw; iw = i; wall = w; }
        } else di = dot3(nb, V + 3 * i) + c0;
        if (ia < 0 || di < dmin) { ia = i; dmin = di; }
      }
      const int face = iw >= 0 && dwmin <= margin;      /* emitted after the hull's patch contacts (same order rule as above) */
#define HULL_FACE() do { if (face) { real pwf[3]; mat_vec(pwf, R, V + 3 * iw); for (int q = 0; q < 3; q++) pwf[q] += xp[q]; \
                                     add_contact(m, d, g, dwmin, pwf, kWallNormal[wall - 1]); } } while (0)
      if (ia < 0 || dmin > margin) { HULL_FACE(); continue; }
      real thr = dmin + m->hull_skin; if (thr > margin) thr = margin;
      int sel[4] = {ia, -1, -1, -1}; int nsel = 1;
      const real* va = V + 3 * ia;
      /* b: farthest candidate from a */
      int ib = -1; real best = (real)1e-10;
      for (int i = 0; i < nvv; i++) {
        real di = VDIST(i); if (di > thr) continue;
        real e[3] = {V[3 * i] - va[0], V[3 * i + 1] - va[1], V[3 * i + 2] - va[2]};
        real s = dot3(e, e); if (s > best) { best = s; ib = i; }
      }
      if (ib >= 0) {
        sel[nsel++] = ib;
        const real* vb = V + 3 * ib;
        real ab[3] = {vb[0] - va[0], vb[1] - va[1], vb[2] - va[2]};
        real lab2 = dot3(ab, ab);
        /* c: farthest from line ab */
        int ic = -1; best = (real)1e-10 * lab2; real side_c = 0;
        for (int i = 0; i < nvv; i++) {
          real di = VDIST(i); if (di > thr) continue;
          real e[3] = {V[3 * i] - va[0], V[3 * i + 1] - va[1], V[3 * i + 2] - va[2]};
          real cr[3]; cross3(cr, e, ab);
          real s = dot3(cr, cr); if (s > best) { best = s; ic = i; side_c = dot3(cr, nb); }
        }
        if (ic >= 0) {
          sel[nsel++] = ic;
          /* d: farthest on the other side of line ab */
          int id = -1; real sg = side_c > 0 ? (real)-1 : (real)1;
          best = R_SQRT((real)1e-10 * lab2);
          for (int i = 0; i < nvv; i++) {
            real di = VDIST(i); if (di > thr) continue;
            real e[3] = {V[3 * i] - va[0], V[3 * i + 1] - va[1], V[3 * i + 2] - va[2]};
            real cr[3]; cross3(cr, e, ab);
            real s = sg * dot3(cr, nb); if (s > best) { best = s; id = i; }
          }
          if (id >= 0) sel[nsel++] = id;
        }
      }
      if (nsel > m->sem_max_hull_contacts) nsel = m->sem_max_hull_contacts;
      for (int k = 0; k < nsel; k++) {
        const real* v = V + 3 * sel[k];
        real dist = VDIST(sel[k]);
        real pw[3]; mat_vec(pw, R, v);
        for (int q = 0; q < 3; q++) pw[q] += xp[q];
        add_contact(m, d, g, dist, pw, n);
      }
      HULL_FACE();
#undef HULL_FACE
#undef VDIST
    }
  }
}

/* ------------------------------------------------------------------ stage: constraint rows */
static void point_jac_dir(const omodel* m, const odata* d, int body, const real* p, const real* dir, real* row) {
  /* row[j] = dir · (v_j + w_j × (p − o)) for dofs on the path body→root, else 0 */
  const real* o = d->xpos;
  real r[3] = {p[0] - o[0], p[1] - o[1], p[2] - o[2]};
  memset(row, 0, sizeof(real) * (size_t)m->nv);
  int b = body;
  while (b >= 0) {
    int adr = m->body_dofadr[b];
    for (int j = adr; j < adr + m->body_dofnum[b]; j++) {
      const real* S = d->S + 6 * j; real wr[3]; cross3(wr, S, r);
      row[j] = dir[0] * (S[3] + wr[0]) + dir[1] * (S[4] + wr[1]) + dir[2] * (S[5] + wr[2]);
    }
    b = m->body_parent[b];
  }
}

static real impedance(const real* solimp, real r) {
  /* d(r) of MuJoCo's getimpedance; r = pos − margin */
  real d0 = solimp[0], dmax = solimp[1], width = solimp[2], mid = solimp[3], power = solimp[4];
  if (d0 == dmax || width <= (real)NMF_MINVAL) return (real)0.5 * (d0 + dmax);
  real x = R_FABS(r) / width, y;
  if (x >= 1) y = 1;
  else if (x <= 0) y = 0;
  else if (power == 1) y = x;
  else if (x <= mid) y = R_POW(x, power) / R_POW(mid, power - 1);
  else y = 1 - R_POW(1 - x, power) / R_POW(1 - mid, power - 1);
  return d0 + y * (dmax - d0);
}

static void make_constraints(const omodel* m, odata* d) {
  int nv = m->nv;
  d->nefc = 4 * d->ncon;
  for (int c = 0; c < d->ncon; c++) {
    int g = d->con_geom[c], b = m->geom_body[g];
    real* Jn = d->Jc + (size_t)(3 * c) * nv; real* Jt1 = Jn + nv; real* Jt2 = Jt1 + nv;
    point_jac_dir(m, d, b, d->con_pos[c], d->con_frame[c], Jn);
    point_jac_dir(m, d, b, d->con_pos[c], d->con_frame[c] + 3, Jt1);
    point_jac_dir(m, d, b, d->con_pos[c], d->con_frame[c] + 6, Jt2);
    real mu = d->con_mu[c];
    for (int k = 0; k < 4; k++) {
      real* row = d->J + (size_t)(4 * c + k) * nv; const real* Jt = k < 2 ? Jt1 : Jt2;
      real sg = (k & 1) ? -mu : mu;
      for (int j = 0; j < nv; j++) row[j] = Jn[j] + sg * Jt[j];
    }
    /* impedance, regulariser, reference acceleration parameters */
    const real* solref = m->pair_solref + 2 * g; const real* solimp = m->pair_solimp + 5 * g;
    real margin = m->pair_margin[g];
    real r = d->con_dist[c] - margin;
    real imp = impedance(solimp, r);
    real tran = m->geom_invweight0[g];
    real diagA = tran + mu * mu * tran;                 /* pyramidal edge: (1 + mu²)·tran */
    real Rn = ((real)1 - imp) * diagA / imp; if (Rn < (real)NMF_MINVAL) Rn = (real)NMF_MINVAL;
    real Rpy = m->sem_pyramid_plain ? Rn : 2 * mu * mu * Rn;   /* all edges of the pyramid share R */
    if (Rpy < (real)NMF_MINVAL) Rpy = (real)NMF_MINVAL;
    real tc = solref[0], dr = solref[1], K, B;
    if (tc > 0) {
      if (tc < 2 * m->timestep) tc = 2 * m->timestep;  /* refsafe */
      real dmax = solimp[1];
      K = (real)1 / (dmax * dmax * tc * tc * dr * dr);
      B = (real)2 / (dmax * tc);
    } else { K = -tc / (solimp[1] * solimp[1]); B = -dr / solimp[1]; }
    for (int k = 0; k < 4; k++) {
      int i = 4 * c + k;
      d->efc_R[i] = Rpy; d->efc_D[i] = (real)1 / Rpy;
      d->efc_KBI[i][0] = K; d->efc_KBI[i][1] = B; d->efc_KBI[i][2] = imp;
      d->efc_pos[i] = r; d->efc_bilateral[i] = 0;
    }
  }
  if (m->weld_active) {
    /* six bilateral rows = the components (w; v) of the root twist; residual = pose error of the root body */
    real qt[4] = {m->weld_quat[0], -m->weld_quat[1], -m->weld_quat[2], -m->weld_quat[3]}, qe[4];
    quat_mul(qe, d->xquat, qt);
    real sg = qe[0] < 0 ? (real)-2 : (real)2;
    real res[6] = {sg * qe[1], sg * qe[2], sg * qe[3],
                   d->xpos[0] - m->weld_pos[0], d->xpos[1] - m->weld_pos[1], d->xpos[2] - m->weld_pos[2]};
    for (int r = 0; r < 6; r++) {
      int i = d->nefc + r;
      real* row = d->J + (size_t)i * nv;
      memset(row, 0, sizeof(real) * (size_t)nv);
      for (int j = 0; j < 6; j++) row[j] = d->S[6 * j + r];
      real imp = impedance(m->weld_solimp, res[r]);
      real dA = m->weld_invweight[r < 3 ? 1 : 0];
      real R = ((real)1 - imp) * dA / imp; if (R < (real)NMF_MINVAL) R = (real)NMF_MINVAL;
      real tc = m->weld_solref[0], dr = m->weld_solref[1], K, B;
      if (tc > 0) { if (tc < 2 * m->timestep) tc = 2 * m->timestep;
        K = (real)1 / (m->weld_solimp[1] * m->weld_solimp[1] * tc * tc * dr * dr); B = (real)2 / (m->weld_solimp[1] * tc);
      } else { K = -tc / (m->weld_solimp[1] * m->weld_solimp[1]); B = -dr / m->weld_solimp[1]; }
      d->efc_R[i] = R; d->efc_D[i] = (real)1 / R;
      d->efc_KBI[i][0] = K; d->efc_KBI[i][1] = B; d->efc_KBI[i][2] = imp;
      d->efc_pos[i] = res[r]; d->efc_bilateral[i] = 1;
    }
    d->nefc += 6;
  }
}

/* ------------------------------------------------------------------ stage: velocity, bias */
static void velocity_and_bias(const omodel* m, odata* d) {
  for (int b = 0; b < m->nb; b++) {
    real* v = d->cvel + 6 * b; real* a = d->cacc + 6 * b; int p = m->body_parent[b];
    if (p < 0) { for (int k = 0; k < 6; k++) v[k] = 0; a[0] = a[1] = a[2] = 0;
                 a[3] = -m->gravity[0]; a[4] = -m->gravity[1]; a[5] = -m->gravity[2]; }
    else { memcpy(v, d->cvel + 6 * p, 6 * sizeof(real)); memcpy(a, d->cacc + 6 * p, 6 * sizeof(real)); }
    int adr = m->body_dofadr[b], num = m->body_dofnum[b];
    if (p < 0) {
      for (int j = 0; j < 3; j++) for (int k = 0; k < 6; k++) v[k] += d->S[6 * j + k] * d->qvel[j];
      real vt[6]; memcpy(vt, v, sizeof(vt));          /* all three rotational Ṡ use the same velocity */
      for (int j = 3; j < 6; j++) {
        real sd[6]; cross_motion(sd, vt, d->S + 6 * j);
        for (int k = 0; k < 6; k++) { a[k] += sd[k] * d->qvel[j]; v[k] += d->S[6 * j + k] * d->qvel[j]; }
      }
    } else {
      for (int j = adr; j < adr + num; j++) {
        real sd[6]; cross_motion(sd, v, d->S + 6 * j);
        for (int k = 0; k < 6; k++) { a[k] += sd[k] * d->qvel[j]; v[k] += d->S[6 * j + k] * d->qvel[j]; }
      }
    }
    real Iv[6], Ia[6], vxIv[6];
    inert_mul(Iv, d->Ib + 10 * b, v); inert_mul(Ia, d->Ib + 10 * b, a); cross_force(vxIv, v, Iv);
    for (int k = 0; k < 6; k++) d->cfrc[6 * b + k] = Ia[k] + vxIv[k];
  }
  for (int b = m->nb - 1; b > 0; b--) {
    int p = m->body_parent[b];
    for (int k = 0; k < 6; k++) d->cfrc[6 * p + k] += d->cfrc[6 * b + k];
  }
  for (int j = 0; j < m->nv; j++) d->qfrc_bias[j] = dot6(d->S + 6 * j, d->cfrc + 6 * m->dof_body[j]);
  for (int j = 0; j < m->nv; j++) {
    real q = j < 6 ? 0 : d->qpos[j + 1];
    d->qfrc_passive[j] = j < 6 ? 0 : -m->dof_stiffness[j] * (q - m->dof_springref[j]) - m->dof_damping[j] * d->qvel[j];
  }
}

/* ------------------------------------------------------------------ stage: actuation */
#define NMF_ACTGEN 32
static inline real rmax(real a, real b) { return a > b ? a : b; }
static inline real rclip(real x, real lo, real hi) { return x < lo ? lo : (x > hi ? hi : x); }
/* MuJoCo's muscle model as its documentation states it (computation/index.html#muscle-actuators, engine_util_misc.c
 * mju_muscleGain / mju_muscleBias / mju_muscleDynamics; mujoco 3.6.0, absent from this image: restated, not linked).
 * prm: range0 range1 force scale lmin lmax vmax fpmax fvmax */
static real muscle_length(real len, const real* lr, const real* prm, real* L0out) {
  real L0 = (lr[1] - lr[0]) / rmax((real)NMF_MINVAL, prm[1] - prm[0]);
  if (L0out) *L0out = L0;
  return prm[0] + (len - lr[0]) / rmax((real)NMF_MINVAL, L0);
}
static real muscle_peak(const real* prm, real acc0) { return prm[2] < 0 ? prm[3] / rmax((real)NMF_MINVAL, acc0) : prm[2]; }
static real muscle_gain(real len, real vel, const real* lr, real acc0, const real* prm) {
  real lmin = prm[4], lmax = prm[5], vmax = prm[6], fvmax = prm[8], L0;
  real L = muscle_length(len, lr, prm, &L0);
  real V = vel / rmax((real)NMF_MINVAL, L0 * vmax);
  real a = (real)0.5 * (lmin + 1), b = (real)0.5 * (1 + lmax), FL = 0, FV, x;
  if (L >= lmin && L <= a) { x = (L - lmin) / rmax((real)NMF_MINVAL, a - lmin); FL = (real)0.5 * x * x; }
  else if (L > a && L <= 1) { x = (1 - L) / rmax((real)NMF_MINVAL, 1 - a); FL = 1 - (real)0.5 * x * x; }
  else if (L > 1 && L <= b) { x = (L - 1) / rmax((real)NMF_MINVAL, b - 1); FL = 1 - (real)0.5 * x * x; }
  else if (L > b && L <= lmax) { x = (lmax - L) / rmax((real)NMF_MINVAL, lmax - b); FL = (real)0.5 * x * x; }
  real y = fvmax - 1;
  if (V <= -1) FV = 0;
  else if (V <= 0) FV = (V + 1) * (V + 1);
  else if (V <= y) FV = fvmax - (y - V) * (y - V) / rmax((real)NMF_MINVAL, y);
  else FV = fvmax;
  return -muscle_peak(prm, acc0) * FL * FV;
}
static real muscle_bias(real len, const real* lr, real acc0, const real* prm) {
  real lmax = prm[5], fpmax = prm[7];
  real L = muscle_length(len, lr, prm, NULL);
  real b = (real)0.5 * (1 + lmax), FP, x;
  if (L <= 1) FP = 0;
  else if (L <= b) { x = (L - 1) / rmax((real)NMF_MINVAL, b - 1); FP = fpmax * (real)0.5 * x * x; }
  else { x = (L - b) / rmax((real)NMF_MINVAL, b - 1); FP = fpmax * ((real)0.5 + x); }
  return -muscle_peak(prm, acc0) * FP;
}
static real muscle_dynamics(real ctrl, real act, const real* prm) {      /* prm: tau_act tau_deact tausmooth */
  real cc = rclip(ctrl, 0, 1), ac = rclip(act, 0, 1);
  real tau_act = prm[0] * ((real)0.5 + (real)1.5 * ac), tau_deact = prm[1] / ((real)0.5 + (real)1.5 * ac);
  real dctrl = cc - act, tau;
  if (prm[2] < (real)NMF_MINVAL) tau = dctrl > 0 ? tau_act : tau_deact;
  else {          /* quintic sigmoid over the smoothing width */
    real x = dctrl / prm[2] + (real)0.5, sg = x <= 0 ? 0 : (x >= 1 ? 1 : x * x * x * (3 * x * (2 * x - 5) + 10));
    tau = tau_deact + (tau_act - tau_deact) * sg;
  }
  return dctrl / rmax((real)NMF_MINVAL, tau);
}
/* one general actuator on hinge dof j (mj_fwdActuation: activation derivative, then force = gain * input + bias, clamped;
 * mj_advance / mj_nextActivation: the next activation, clamped to actrange) — the force uses the activation at the START of
 * the step (option actearly off, MuJoCo's default) */
static real general_actuator(const omodel* m, odata* d, int u, real ctrl) {
  const real* g = m->act_general + (size_t)u * NMF_ACTGEN;
  int j = m->act_trn[u], dyn = (int)g[1], gt = (int)g[2], bt = (int)g[3];
  real gear = g[5], len = gear * d->qpos[j + 1], vel = gear * d->qvel[j], act = d->act[u], h = m->timestep;
  const real *dynprm = g + 6, *gainprm = g + 9, *biasprm = g + 18, *lr = g + 29; real acc0 = g[31];
  real act_dot = 0;
  if (dyn == 1) act_dot = ctrl;
  else if (dyn == 2 || dyn == 3) act_dot = (ctrl - act) / rmax((real)NMF_MINVAL, dynprm[0]);
  else if (dyn == 4) act_dot = muscle_dynamics(ctrl, act, dynprm);
  if (dyn) {
    real nx;
    if (dyn == 3) { real tau = rmax((real)NMF_MINVAL, dynprm[0]); nx = act + act_dot * tau * (1 - R_EXP(-h / tau)); }
    else nx = act + act_dot * h;
    if (g[4] != 0) nx = rclip(nx, g[27], g[28]);
    d->act_next[u] = nx;
  }
  real input = dyn ? act : ctrl;
  real gain = gt == 0 ? gainprm[0] : gt == 1 ? gainprm[0] + gainprm[1] * len + gainprm[2] * vel : muscle_gain(len, vel, lr, acc0, gainprm);
  real f = gain * input;
  if (bt == 1) f += biasprm[0] + biasprm[1] * len + biasprm[2] * vel;
  else if (bt == 2) f += muscle_bias(len, lr, acc0, biasprm);
  if ((int)g[0] & 2) f = rclip(f, m->act_forcerange[2 * u], m->act_forcerange[2 * u + 1]);
  d->actuator_force[u] = f;
  d->qfrc_actuator[j] += gear * f;
  return f;
}

static void actuation(const omodel* m, odata* d) {
  int nv = m->nv;
  memset(d->qfrc_actuator, 0, sizeof(real) * (size_t)nv);
  for (int u = 0; u < m->nu; u++) {
    real ctrl = d->ctrl[u];
    if (m->act_limited[2 * u + 1]) {
      if (ctrl < m->act_ctrlrange[2 * u]) ctrl = m->act_ctrlrange[2 * u];
      if (ctrl > m->act_ctrlrange[2 * u + 1]) ctrl = m->act_ctrlrange[2 * u + 1];
    }
    real f;
    d->act_next[u] = d->act[u];
    if (m->act_general && m->act_general[(size_t)u * NMF_ACTGEN] != 0) { general_actuator(m, d, u, ctrl); continue; }
    if (m->act_type[u] == ACT_ADHESION) {
      f = m->act_gain[u] * ctrl;
      d->actuator_force[u] = f;
      int body = m->act_trn[u], cnt = 0;
      real* mom = d->act_moment + (size_t)u * nv;
      memset(mom, 0, sizeof(real) * (size_t)nv);
      /* contacts of the adhesion segment's own geom (the MJCF body the actuator names, reference fly.py:434-439);
       * sem_adhesion_fused: every contact of the dynamic body the segment was merged into */
      for (int c = 0; c < d->ncon; c++)
        if (m->sem_adhesion_fused ? m->geom_body[d->con_geom[c]] == body : d->con_geom[c] == m->act_geom[u]) {
        const real* Jn = d->Jc + (size_t)(3 * c) * nv;
        for (int j = 0; j < nv; j++) mom[j] -= Jn[j];
        cnt++;
      }
      if (cnt) for (int j = 0; j < nv; j++) { mom[j] /= (real)cnt; d->qfrc_actuator[j] += mom[j] * f; }
    } else {
      int j = m->act_trn[u];
      real q = d->qpos[j + 1], qd = d->qvel[j];
      f = m->act_gain[u] * ctrl + m->act_bias[2 * u] * q + m->act_bias[2 * u + 1] * qd;
      if (m->act_limited[2 * u]) {
        if (f < m->act_forcerange[2 * u]) f = m->act_forcerange[2 * u];
        if (f > m->act_forcerange[2 * u + 1]) f = m->act_forcerange[2 * u + 1];
      }
      d->actuator_force[u] = f;
      d->qfrc_actuator[j] += f;
    }
  }
}

/* ------------------------------------------------------------------ stage: Newton solve */
static real constraint_cost_b(int nefc, const real* D, const real* jar, const int* bil) {
  real c = 0;
  for (int i = 0; i < nefc; i++) if (bil[i] || jar[i] < 0) c += (real)0.5 * D[i] * jar[i] * jar[i];
  return c;
}

static void solve_constraints(const omodel* m, odata* d) {
  int nv = m->nv, nefc = d->nefc;
  real* qacc = d->qacc; real* Ma = d->w1; real* grad = d->w2; real* search = d->w3; real* Mv = d->w4;
  real* jar = d->w5; /* nefc */
  real jv[NMF_MAXEFC];
  d->solver_iter = 0;
  memset(d->qfrc_constraint, 0, sizeof(real) * (size_t)nv);
  memset(d->efc_force, 0, sizeof(d->efc_force));
  if (nefc == 0) { memcpy(qacc, d->qacc_smooth, sizeof(real) * (size_t)nv); d->solver_cost = 0; return; }
  /* reference acceleration */
  for (int i = 0; i < nefc; i++) {
    const real* row = d->J + (size_t)i * nv; real vel = 0;
    for (int j = 0; j < nv; j++) vel += row[j] * d->qvel[j];
    d->efc_aref[i] = -d->efc_KBI[i][1] * vel - d->efc_KBI[i][0] * d->efc_KBI[i][2] * d->efc_pos[i];
  }
  /* warm start: pick the cheaper of qacc_warmstart and qacc_smooth */
  real cost_ws, cost_sm;
  {
    memcpy(qacc, d->qacc_warmstart, sizeof(real) * (size_t)nv);
    mul_M(m, d->M, qacc, Ma);
    real g = 0; for (int j = 0; j < nv; j++) g += (real)0.5 * (qacc[j] - d->qacc_smooth[j]) * (Ma[j] - d->qfrc_smooth[j]);
    for (int i = 0; i < nefc; i++) { real s = 0; const real* row = d->J + (size_t)i * nv;
      for (int j = 0; j < nv; j++) s += row[j] * qacc[j]; jar[i] = s - d->efc_aref[i]; }
    cost_ws = g + constraint_cost_b(nefc, d->efc_D, jar, d->efc_bilateral);
    real js[NMF_MAXEFC];
    for (int i = 0; i < nefc; i++) { real s = 0; const real* row = d->J + (size_t)i * nv;
      for (int j = 0; j < nv; j++) s += row[j] * d->qacc_smooth[j]; js[i] = s - d->efc_aref[i]; }
    cost_sm = constraint_cost_b(nefc, d->efc_D, js, d->efc_bilateral);
    if (cost_sm < cost_ws) {
      memcpy(qacc, d->qacc_smooth, sizeof(real) * (size_t)nv);
      memcpy(Ma, d->qfrc_smooth, sizeof(real) * (size_t)nv);
      memcpy(jar, js, sizeof(real) * (size_t)nefc);
      cost_ws = cost_sm;
    }
  }
  real cost = cost_ws;
  real scale = (real)1 / (m->meaninertia * (real)(nv > 1 ? nv : 1));
  for (int iter = 0; iter < m->max_iter; iter++) {
    /* gradient and Hessian */
    for (int j = 0; j < nv; j++) grad[j] = Ma[j] - d->qfrc_smooth[j];
    memcpy(d->H, d->M, sizeof(real) * (size_t)nv * nv);
    for (int i = 0; i < nefc; i++) if (d->efc_bilateral[i] || jar[i] < 0) {
      const real* row = d->J + (size_t)i * nv; real Di = d->efc_D[i]; real f = -Di * jar[i];
      for (int j = 0; j < nv; j++) if (row[j] != 0) {
        grad[j] -= row[j] * f;
        real s = Di * row[j];
        for (int k = 0; k <= j; k++) if (row[k] != 0) { d->H[j * nv + k] += s * row[k]; if (k != j) d->H[k * nv + j] += s * row[k]; }
      }
    }
    /* stop when the gradient is below the tolerance OR at its own rounding-noise level:
       grad is a sum of three vectors, so its noise floor is ~eps * |magnitudes| */
    real gn = 0, gm = 0;
    for (int j = 0; j < nv; j++) {
      gn += grad[j] * grad[j];
      real jtf = grad[j] - (Ma[j] - d->qfrc_smooth[j]);
      real mag = R_FABS(Ma[j]) + R_FABS(d->qfrc_smooth[j]) + R_FABS(jtf);
      gm += mag * mag;
    }
    if (scale * R_SQRT(gn) < m->tolerance) break;
    if (d->solver_mode == 0 && R_SQRT(gn) <= NMF_NOISE_FACTOR * R_EPS * R_SQRT(gm)) break;
    factor_tree(m, d->H, d->L, d->Ld);
    for (int j = 0; j < nv; j++) search[j] = -grad[j];
    solve_tree(m, d->L, d->Ld, search);
    /* exact line search on the convex piecewise quadratic φ(α) */
    mul_M(m, d->M, search, Mv);
    real g1 = 0, g2 = 0;
    for (int j = 0; j < nv; j++) { g1 += search[j] * (Ma[j] - d->qfrc_smooth[j]); g2 += search[j] * Mv[j]; }
    for (int i = 0; i < nefc; i++) { real s = 0; const real* row = d->J + (size_t)i * nv;
      for (int j = 0; j < nv; j++) s += row[j] * search[j]; jv[i] = s; }
    real alpha = 0, lo = 0, hi = -1; /* hi < 0: no upper bracket yet */
    for (int ls = 0; ls < 30; ls++) {
      real d1 = g1 + alpha * g2, d2 = g2;
      for (int i = 0; i < nefc; i++) { real x = jar[i] + alpha * jv[i];
        if (d->efc_bilateral[i] || x < 0) { d1 += d->efc_D[i] * x * jv[i]; d2 += d->efc_D[i] * jv[i] * jv[i]; } }
      if (d2 <= 0 || d1 == 0) break;
      if (d1 < 0) lo = alpha; else hi = alpha;
      real next = alpha - d1 / d2;
      int bisected = 0;
      if (hi >= 0 && (next <= lo || next >= hi)) { next = (real)0.5 * (lo + hi); bisected = 1; }
      /* phi' is linear while the active set does not change: then `next` is the exact minimiser */
      int same = !bisected;
      if (d->solver_mode != 0) same = 0;
      for (int i = 0; same && i < nefc; i++)
        same = d->efc_bilateral[i] || (((jar[i] + alpha * jv[i]) < 0) == ((jar[i] + next * jv[i]) < 0));
      real change = R_FABS(next - alpha);
      alpha = next;
      if (same || change <= (real)8 * R_EPS * R_FABS(next)) break;
    }
    if (alpha <= 0) break;
    for (int j = 0; j < nv; j++) { qacc[j] += alpha * search[j]; Ma[j] += alpha * Mv[j]; }
    for (int i = 0; i < nefc; i++) jar[i] += alpha * jv[i];
    real gq = 0; for (int j = 0; j < nv; j++) gq += (real)0.5 * (qacc[j] - d->qacc_smooth[j]) * (Ma[j] - d->qfrc_smooth[j]);
    real newcost = gq + constraint_cost_b(nefc, d->efc_D, jar, d->efc_bilateral);
    d->solver_iter = iter + 1;
    real improvement = cost - newcost;
    cost = newcost;
    if (scale * improvement < m->tolerance) break;
    /* (shared with the kernel: the rounding-floor test on the improvement only guards against cycling, from the ninth
       iteration on — applied from the first it stopped float32 solves that were still moving, round 4) */
    if (d->solver_mode == 0 && iter >= 8 && improvement <= NMF_NOISE_FACTOR * R_EPS * R_FABS(cost)) break;
  }
  d->solver_cost = cost;
  for (int i = 0; i < nefc; i++) {
    real f = (d->efc_bilateral[i] || jar[i] < 0) ? -d->efc_D[i] * jar[i] : 0;
    d->efc_force[i] = f;
    if (f != 0) { const real* row = d->J + (size_t)i * nv; for (int j = 0; j < nv; j++) d->qfrc_constraint[j] += row[j] * f; }
  }
}

/* ------------------------------------------------------------------ stage: noslip post-pass (CPU flavour only)
 * MuJoCo's mj_solNoSlip as documented (computation chapter, "Noslip solver"; the reference's CPU class runs it with
 * option/noslip_iterations = 5, src/flygym/assets/model/mujoco_globals.yaml:15, src/flygym/simulation.py:74-76; the batched
 * class strips it, src/flygym/warp/simulation.py:427-448): a Gauss-Seidel pass over the FRICTION dimensions of the contacts
 * with the regulariser removed (A = J M^-1 J^T, no R).  Pyramidal cones: the normal force of a contact is the sum of its
 * pyramid-edge forces, so every pair of opposing edges (f0, f1) = (mid + y, mid - y) keeps its sum and y in [-mid, mid]
 * minimises 1/2 f^T A f + f^T b (b = J qacc_smooth - aref); a pair update that raises the cost is undone; the pass stops after
 * noslip_iterations sweeps or when a sweep improves the cost by less than noslip_tolerance (1e-6, MuJoCo's default), scaled by
 * 1 / (meaninertia nv).  Then qfrc_constraint = J^T f and qacc = M^-1 (qfrc_smooth + qfrc_constraint). */
#define NMF_NOSLIP_TOLERANCE ((real)1e-6)
static void noslip(const omodel* m, odata* d) {
  int nv = m->nv, nefc = d->nefc;
  if (!d->noslip_on || m->noslip_iter <= 0 || d->ncon == 0) return;
  real* A = (real*)malloc(sizeof(real) * (size_t)nefc * nefc);
  real* b = (real*)malloc(sizeof(real) * (size_t)nefc);
  real* col = (real*)malloc(sizeof(real) * (size_t)nv);
  factor_tree(m, d->M, d->L, d->Ld);
  for (int i = 0; i < nefc; i++) {
    memcpy(col, d->J + (size_t)i * nv, sizeof(real) * (size_t)nv);
    solve_tree(m, d->L, d->Ld, col);
    for (int k = 0; k < nefc; k++) { real s = 0; const real* row = d->J + (size_t)k * nv; for (int j = 0; j < nv; j++) s += row[j] * col[j]; A[(size_t)k * nefc + i] = s; }
    real s = 0; const real* row = d->J + (size_t)i * nv; for (int j = 0; j < nv; j++) s += row[j] * d->qacc_smooth[j];
    b[i] = s - d->efc_aref[i];
  }
  real* f = d->efc_force;
  real scale = (real)1 / (m->meaninertia * (real)(nv > 1 ? nv : 1));
  for (int iter = 0; iter < m->noslip_iter; iter++) {
    real improvement = 0;
    if (iter == 0) for (int i = 0; i < nefc; i++) improvement += (real)0.5 * f[i] * f[i] / d->efc_D[i];   /* the regulariser's share of the cost drops out */
    for (int c = 0; c < d->ncon; c++) {                              /* contact c owns rows 4c .. 4c + 3; the weld's rows follow the contacts' */
      int i = 4 * c;
      for (int p = 0; p < 2; p++) {
        int j0 = i + 2 * p, j1 = j0 + 1;
        real old0 = f[j0], old1 = f[j1], res0 = b[j0], res1 = b[j1];
        for (int k = 0; k < nefc; k++) { res0 += A[(size_t)j0 * nefc + k] * f[k]; res1 += A[(size_t)j1 * nefc + k] * f[k]; }
        real a00 = A[(size_t)j0 * nefc + j0], a01 = A[(size_t)j0 * nefc + j1], a11 = A[(size_t)j1 * nefc + j1];
        real bc0 = res0 - a00 * old0 - a01 * old1, bc1 = res1 - a01 * old0 - a11 * old1;
        real mid = (real)0.5 * (old0 + old1);
        real K1 = a00 + a11 - (real)2 * a01, K0 = mid * (a00 - a11) + bc0 - bc1;
        real n0, n1;
        if (K1 < (real)1e-15) { n0 = mid; n1 = mid; }
        else { real y = -K0 / K1; if (y < -mid) y = -mid; if (y > mid) y = mid; n0 = mid + y; n1 = mid - y; }
        real d0 = n0 - old0, d1 = n1 - old1;
        real change = (real)0.5 * (d0 * (a00 * d0 + a01 * d1) + d1 * (a01 * d0 + a11 * d1)) + d0 * res0 + d1 * res1;
        if (change > (real)1e-10) { n0 = old0; n1 = old1; change = 0; }
        f[j0] = n0; f[j1] = n1;
        improvement -= change;
      }
    }
    if (scale * improvement < NMF_NOSLIP_TOLERANCE) break;
  }
  memset(d->qfrc_constraint, 0, sizeof(real) * (size_t)nv);
  for (int i = 0; i < nefc; i++) if (f[i] != 0) { const real* row = d->J + (size_t)i * nv; for (int j = 0; j < nv; j++) d->qfrc_constraint[j] += row[j] * f[i]; }
  for (int j = 0; j < nv; j++) d->qacc[j] = d->qfrc_smooth[j] + d->qfrc_constraint[j];
  solve_tree(m, d->L, d->Ld, d->qacc);
  free(A); free(b); free(col);
}

/* ------------------------------------------------------------------ stage: contact sensors */
static void contact_sensors(const omodel* m, odata* d) {
  memset(d->sensordata, 0, sizeof(real) * 96);
  if (!m->nsensor) return;
  for (int leg = 0; leg < 6; leg++) {
    real* out = d->sensordata + 16 * leg;
    real F[3] = {0, 0, 0}, wsum = 0, pc[3] = {0, 0, 0}, pm[3] = {0, 0, 0}; int cnt = 0, first = -1;
    for (int c = 0; c < d->ncon; c++) {
      if (m->geom_sensor[d->con_geom[c]] != leg) continue;
      const real* f = d->efc_force + 4 * c; real mu = d->con_mu[c];
      real fn = f[0] + f[1] + f[2] + f[3];
      if (first < 0) first = c;
      cnt++;
      wsum += fn;
      for (int k = 0; k < 3; k++) { pc[k] += fn * d->con_pos[c][k]; pm[k] += d->con_pos[c][k]; }
      (void)mu;
    }
    if (!cnt) continue;
    for (int k = 0; k < 3; k++) pc[k] = wsum > 0 ? pc[k] / wsum : pm[k] / (real)cnt;
    real T[3] = {0, 0, 0};
    for (int c = 0; c < d->ncon; c++) {
      if (m->geom_sensor[d->con_geom[c]] != leg) continue;
      const real* f = d->efc_force + 4 * c; real mu = d->con_mu[c]; const real* fr = d->con_frame[c];
      real fn = f[0] + f[1] + f[2] + f[3], ft1 = mu * (f[0] - f[1]), ft2 = mu * (f[2] - f[3]);
      real Fc[3], r[3], t[3];
      for (int k = 0; k < 3; k++) { Fc[k] = fn * fr[k] + ft1 * fr[3 + k] + ft2 * fr[6 + k]; r[k] = d->con_pos[c][k] - pc[k]; F[k] += Fc[k]; }
      cross3(t, r, Fc);
      for (int k = 0; k < 3; k++) T[k] += t[k];
    }
    out[0] = (real)cnt;
    if (m->sem_sensor_contact_frame) {   /* net force / torque expressed in the contact frame (normal, t1, t2) */
      const real* fr = d->con_frame[first];
      real Fl[3] = {dot3(fr, F), dot3(fr + 3, F), dot3(fr + 6, F)}, Tl[3] = {dot3(fr, T), dot3(fr + 3, T), dot3(fr + 6, T)};
      memcpy(F, Fl, sizeof(F)); memcpy(T, Tl, sizeof(T));
    }
    for (int k = 0; k < 3; k++) { out[1 + k] = F[k]; out[4 + k] = T[k]; out[7 + k] = pc[k];
      out[10 + k] = d->con_frame[first][k]; out[13 + k] = d->con_frame[first][3 + k]; }
  }
}

/* ------------------------------------------------------------------ forward + integrate */
EXPORT void SFX(nmfo_forward)(const void* mv, void* dv) {
  const omodel* m = (const omodel*)mv; odata* d = (odata*)dv;
  int nv = m->nv;
  kinematics(m, d);
  named_poses(m, d);
  body_inertias(m, d);
  crba(m, d);
  factor_tree(m, d->M, d->L, d->Ld);
  collide(m, d);
  make_constraints(m, d);
  velocity_and_bias(m, d);
  actuation(m, d);
  for (int j = 0; j < nv; j++) d->qfrc_smooth[j] = d->qfrc_passive[j] - d->qfrc_bias[j] + d->qfrc_actuator[j];
  memcpy(d->qacc_smooth, d->qfrc_smooth, sizeof(real) * (size_t)nv);
  solve_tree(m, d->L, d->Ld, d->qacc_smooth);
  solve_constraints(m, d);
  /* the next step's warm start is the main solver's result: MuJoCo saves it before the noslip post-pass (mj_fwdConstraint) */
  memcpy(d->qacc_warmstart, d->qacc, sizeof(real) * (size_t)nv);
  noslip(m, d);
  contact_sensors(m, d);
}

static void integrate(const omodel* m, odata* d) {
  int nv = m->nv; real h = m->timestep;
  /* implicit joint damping: (M + h·diag(B)) a = qfrc_smooth + qfrc_constraint */
  real* rhs = d->w1;
  memcpy(d->H, d->M, sizeof(real) * (size_t)nv * nv);
  for (int j = 0; j < nv; j++) { d->H[j * nv + j] += h * m->dof_damping[j]; rhs[j] = d->qfrc_smooth[j] + d->qfrc_constraint[j]; }
  factor_tree(m, d->H, d->L, d->Ld);
  solve_tree(m, d->L, d->Ld, rhs);
  for (int j = 0; j < nv; j++) d->qvel[j] += h * rhs[j];
  for (int k = 0; k < 3; k++) d->qpos[k] += h * d->qvel[k];
  real w[3] = {d->qvel[3], d->qvel[4], d->qvel[5]};
  real wn = R_SQRT(dot3(w, w));
  if (wn > (real)NMF_MINVAL) {
    real ax[3] = {w[0] / wn, w[1] / wn, w[2] / wn}, dq[4], t[4];
    axis_angle_quat(dq, ax, h * wn);
    quat_mul(t, d->qpos + 3, dq);
    memcpy(d->qpos + 3, t, 4 * sizeof(real));
  }
  quat_norm(d->qpos + 3);
  for (int j = 6; j < nv; j++) d->qpos[j + 1] += h * d->qvel[j];
  memcpy(d->act, d->act_next, sizeof(real) * (size_t)m->nu);      /* mj_advance: activations */
  d->time += h;
}

EXPORT void SFX(nmfo_step)(const void* mv, void* dv, int nsteps) {
  for (int s = 0; s < nsteps; s++) { SFX(nmfo_forward)(mv, dv); integrate((const omodel*)mv, (odata*)dv); }
}

/* kinematic replay on the CPU: ctrl[act_ids[a]] = table[(start + s) % table_steps][a] before step s
 * (the reference benchmark loop, src/flygym_demo/benchmark/time_gpu_simulation.py:137-150) */
EXPORT void SFX(nmfo_step_replay)(const void* mv, void* dv, const float* table, int table_steps, int n_act,
                                  const int* act_ids, int start, int nsteps) {
  odata* d = (odata*)dv;
  for (int s = 0; s < nsteps; s++) {
    const float* row = table + (size_t)((start + s) % table_steps) * n_act;
    for (int a = 0; a < n_act; a++) d->ctrl[act_ids[a]] = (real)row[a];
    SFX(nmfo_forward)(mv, dv);
    integrate((const omodel*)mv, d);
  }
}

EXPORT void SFX(nmfo_reset)(const void* mv, void* dv) {
  const omodel* m = (const omodel*)mv; odata* d = (odata*)dv;
  memcpy(d->qpos, m->key_qpos, sizeof(real) * (size_t)m->nq);
  memset(d->qvel, 0, sizeof(real) * (size_t)m->nv);
  memset(d->qacc_warmstart, 0, sizeof(real) * (size_t)m->nv);
  memcpy(d->ctrl, m->key_ctrl, sizeof(real) * (size_t)m->nu);
  memset(d->act, 0, sizeof(real) * (size_t)m->nu); memset(d->act_next, 0, sizeof(real) * (size_t)m->nu);
  memset(d->actuator_force, 0, sizeof(real) * (size_t)m->nu);
  memset(d->sensordata, 0, sizeof(real) * 96);
  d->time = 0; d->ncon = 0; d->nefc = 0; d->overflow = 0; d->solver_iter = 0;
  kinematics(m, d);
  named_poses(m, d);
}

/* ------------------------------------------------------------------ accessors */
EXPORT void* SFX(nmfo_ptr)(const void* mv, void* dv, const char* name, int* count) {
  const omodel* m = (const omodel*)mv; odata* d = (odata*)dv; int nv = m->nv;
#define F(n, p, c) if (strcmp(name, n) == 0) { *count = (c); return (void*)(p); }
  F("qpos", d->qpos, m->nq) F("qvel", d->qvel, nv) F("ctrl", d->ctrl, m->nu) F("act", d->act, m->nu)
  F("qacc_warmstart", d->qacc_warmstart, nv) F("qacc", d->qacc, nv) F("qacc_smooth", d->qacc_smooth, nv)
  F("xpos", d->xpos, 3 * m->nb) F("xquat", d->xquat, 4 * m->nb) F("xmat", d->xmat, 9 * m->nb)
  F("M", d->M, nv * nv) F("qfrc_bias", d->qfrc_bias, nv) F("qfrc_passive", d->qfrc_passive, nv)
  F("qfrc_actuator", d->qfrc_actuator, nv) F("qfrc_smooth", d->qfrc_smooth, nv)
  F("qfrc_constraint", d->qfrc_constraint, nv) F("actuator_force", d->actuator_force, m->nu)
  F("sensordata", d->sensordata, 96) F("seg_xpos", d->seg_xpos, 3 * m->nseg)
  F("seg_xquat", d->seg_xquat, 4 * m->nseg) F("site_xpos", d->site_xpos, 3 * m->nsite)
  F("con_dist", d->con_dist, d->ncon) F("con_pos", d->con_pos, 3 * d->ncon) F("con_frame", d->con_frame, 9 * d->ncon)
  F("efc_force", d->efc_force, d->nefc) F("efc_aref", d->efc_aref, d->nefc) F("efc_D", d->efc_D, d->nefc)
  F("J", d->J, d->nefc * nv) F("cvel", d->cvel, 6 * m->nb) F("S", d->S, 6 * nv) F("time", &d->time, 1)
  F("solver_cost", &d->solver_cost, 1)
#undef F
  *count = 0; return NULL;
}

EXPORT void SFX(nmfo_set_solver_mode)(void* dv, int mode) { ((odata*)dv)->solver_mode = mode; }
EXPORT void SFX(nmfo_set_noslip)(void* dv, int on) { ((odata*)dv)->noslip_on = on; }
EXPORT void SFX(nmfo_set_max_contacts)(void* dv, int n) { ((odata*)dv)->max_contacts = n < 1 ? 1 : (n > NMF_MAXCON ? NMF_MAXCON : n); }

EXPORT void SFX(nmfo_ints)(const void* mv, void* dv, int* out, int* con_geom) {
  (void)mv; odata* d = (odata*)dv;
  out[0] = d->ncon; out[1] = d->nefc; out[2] = d->overflow; out[3] = d->solver_iter;
  if (con_geom) memcpy(con_geom, d->con_geom, sizeof(int) * (size_t)d->ncon);
}
