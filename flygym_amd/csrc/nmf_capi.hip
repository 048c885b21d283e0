// nmf_capi.hip — host side of libnmf_hip.so: the C ABI declared in include/nmf.h.
//
// Owns device memory for the model constants and the per-world state, launches the fused step
// kernel (nmf_step.hip) and the gather/scatter kernels.  No torch types, no host sync on the
// stepping path.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "nmf.h"
#include "nmf_step.hip"
#include "nmf_sensors.hip"
#include "nmf_eyes.hip"
#include "nmf_replay.hip"

namespace {

thread_local std::string g_err;
int fail(const std::string& msg) { g_err = msg; return -1; }

#define HIP_OK(expr)                                                                        \
  do {                                                                                      \
    hipError_t e_ = (expr);                                                                 \
    if (e_ != hipSuccess) return fail(std::string(#expr) + ": " + hipGetErrorString(e_));   \
  } while (0)

// Makes the batch's device current for the calls below and puts the caller's back on every return path: torch (and any
// other HIP user of the thread) reads its current device through hipGetDevice, so a library call must not change it.
struct DeviceGuard {
  int prev = -1;
  bool switched = false;
  hipError_t err = hipSuccess;
  explicit DeviceGuard(int device) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != device) { err = hipSetDevice(device); switched = err == hipSuccess; }
  }
  ~DeviceGuard() { if (switched && prev >= 0) (void)hipSetDevice(prev); }
  DeviceGuard(const DeviceGuard&) = delete;
  DeviceGuard& operator=(const DeviceGuard&) = delete;
};
#define DEVICE_GUARD(b)                                                                       \
  DeviceGuard guard_((b)->device);                                                            \
  if (guard_.err != hipSuccess) return fail(std::string("hipSetDevice: ") + hipGetErrorString(guard_.err))

struct BlobEntry {
  char name[32];
  uint32_t dtype, ndim;
  int64_t shape[4];
  int64_t offset, nbytes;
};

struct HostArray {
  std::vector<float> f;
  std::vector<int32_t> i;
  bool is_int = false;
  int64_t count = 0;
};

}  // namespace

struct nmf_model {
  std::vector<uint8_t> blob;
  std::vector<std::pair<std::string, HostArray>> arrays;
  int nq = 0, nv = 0, nu = 0, nb = 0, nseg = 0, ng = 0, nsite = 0, nsensor = 0, max_iter = 100;
  int star[4] = {0, 0, 0, 0};
  const HostArray* find(const char* name) const {
    for (auto& kv : arrays) if (kv.first == name) return &kv.second;
    return nullptr;
  }
};

// Visit plan of the eye renderer (nmf_eye_plan_create): everything the kernel reads that depends on the id map, the lens and the
// ommatidia types — with its OWN device copies of the id map, the retina run plan, the pale flags and the normalisation, so that
// a render call is pure stream-ordered work whatever the caller does with its buffers afterwards.
struct nmf_eye_plan {
  int device = 0, h = 0, w = 0, n_omm = 0;
  float fov = 0.f;
  int n_groups[3] = {0, 0, 0};                 // [0] chunks that feed an ommatidium, [1] all chunks (frames), [2] sampled mode: pixels
  int* visit[3] = {nullptr, nullptr, nullptr};
  float* cones[3] = {nullptr, nullptr, nullptr};
  float* chunk_cones[3] = {nullptr, nullptr, nullptr};
  int* slot_omm = nullptr;
  int16_t* id_map = nullptr; void* rplan = nullptr; uint8_t* pale = nullptr; float* inv_norm = nullptr;      // the plan's copies
  std::vector<void*> allocs;
  // cache key of nmf_eye_render's implicit plans: the caller's buffer addresses
  const void* key[4] = {nullptr, nullptr, nullptr, nullptr};
};

struct nmf_batch {
  const nmf_model* model = nullptr;
  int n_worlds = 0, device = 0, topo = 0;
  nmf::DevModel dm{};
  nmf::DevModel* dm_dev = nullptr;
  nmf::DevState st{};
  std::vector<void*> allocs;
  float* fields[NMF_FIELD_COUNT] = {};
  int widths[NMF_FIELD_COUNT] = {};
  int64_t steps = 0;
  int* order_buf = nullptr;      // block -> world order of the next stepping launch (see nmf_order_kernel)
  nmf::SchedState* sched_buf = nullptr;
  int resident_waves = 0;        // step-kernel waves the device holds at once
  nmf::ChunkSched* csched_buf = nullptr;   // chunked launches (see nmf_step_kernel): ticket / completion / epoch counters
  unsigned long long* handoff_buf = nullptr;   // chunk hand-off granules (nmf_step_kernel); allocated with the batch
  // eye renderer: the visit plan of the last id map it was called with ([0] chunks that touch an ommatidium, [1] all chunks)
  // nmf_eye_render (the entry without an explicit plan handle) keeps the plans it has built: a few, least recently used first out
  std::vector<nmf_eye_plan*> eye_plans;
  int handoff_stride = 0;
  unsigned long long* clock_probe_buf = nullptr;
  const void* step_fn = nullptr; // the stepping kernel this batch launches (nmf_batch_info)
  int per_cu = 0;                // workgroups of it a CU holds
  unsigned lds_pad = 0;          // idle dynamic LDS per workgroup that holds the residency at options.flies_per_cu (0: none)
  bool chunking = true;          // options.sched = 1 keeps whole-launch work items
  // diagnostics: NMF_SCHED = chunks (default) | plain (= NMF_NO_CHUNKS=1); NMF_ORDER = auto (default) | costliest | inorder |
  // none (no order kernel, worlds in index order) | policy (rounds 1-2: in order or costliest first, whichever measured
  // faster, the other re-tried every 32nd launch).  auto = costliest first for launches of up to 64 steps — the cost of the
  // previous launch predicts this one's (20-step CPG launches 43.1 vs 42.5 M env-steps/s, replay 45.9 vs 45.0 M; 50-step
  // 44.7 vs 44.5 M) — and the measured policy for longer ones, whose costs are a third of a gait cycle stale (250-step
  // launches: 43.7 M costliest first, 45.6 M measured policy)
  int order_every = 1, order_age = 0;      // costliest-first order: recomputed every order_every-th launch (NMF_ORDER_EVERY)
  bool order_valid = false;
  int order_policy = 3;          // 3 auto (default), 1 costliest first, 0 in order, 2 none, -1 the measured policy of rounds 1-2
  int max_chunks = 8, min_chunk_steps = 1;   // NMF_MAX_CHUNKS (<= 16) / NMF_MIN_CHUNK_STEPS / NMF_CHUNK_DIV: tuning experiments
  bool chunk_div_short = false;  // chunk_div applies to launches of up to 64 steps only (longer ones halve)
  double chunk_div = 2.0;        // halving chunks; 1.6 (20 = 13 + 5 + 2, 50 = 32 + 12 + 4 + 2) for the leg-chain kernels on flat ground, see launch()
};

extern "C" const char* nmf_last_error(void) { return g_err.c_str(); }

extern "C" nmf_model* nmf_model_create(const void* blob, size_t nbytes) {
  g_err.clear();
  if (!blob || nbytes < 16 || memcmp(blob, "NMFMODEL", 8) != 0) { fail("nmf_model_create: not an NMFMODEL blob"); return nullptr; }
  auto* m = new nmf_model();
  m->blob.assign((const uint8_t*)blob, (const uint8_t*)blob + nbytes);
  uint32_t version, n;
  memcpy(&version, m->blob.data() + 8, 4);
  memcpy(&n, m->blob.data() + 12, 4);
  if (version != 4) { delete m; fail("nmf_model_create: unsupported blob version (this library reads NMFMODEL v4)"); return nullptr; }
  if ((uint64_t)n > (nbytes - 16) / sizeof(BlobEntry)) { delete m; fail("nmf_model_create: entry table does not fit the blob"); return nullptr; }
  const BlobEntry* e = (const BlobEntry*)(m->blob.data() + 16);
  for (uint32_t k = 0; k < n; ++k) {
    HostArray a;
    int64_t c = 1;
    bool bad = e[k].ndim > 4 || e[k].dtype > 1 || e[k].offset < 0 || e[k].nbytes < 0;
    for (uint32_t d = 0; !bad && d < e[k].ndim; ++d) {
      bad = e[k].shape[d] < 0 || (e[k].shape[d] > 0 && c > (int64_t)nbytes / e[k].shape[d]);
      c *= e[k].shape[d];
    }
    a.count = c;
    const int64_t elem = e[k].dtype == 0 ? 8 : 4;
    if (bad || (uint64_t)e[k].offset > nbytes || (uint64_t)e[k].nbytes > nbytes - (uint64_t)e[k].offset || c * elem > e[k].nbytes) {
      delete m; fail("nmf_model_create: truncated or malformed blob entry"); return nullptr;
    }
    if (e[k].dtype == 0) {
      const double* src = (const double*)(m->blob.data() + e[k].offset);
      a.f.resize((size_t)c);
      for (int64_t i = 0; i < c; ++i) a.f[(size_t)i] = (float)src[i];
    } else {
      a.is_int = true;
      a.i.resize((size_t)c);
      memcpy(a.i.data(), m->blob.data() + e[k].offset, sizeof(int32_t) * (size_t)c);
    }
    char nm[33];
    memcpy(nm, e[k].name, 32); nm[32] = 0;
    m->arrays.emplace_back(std::string(nm), std::move(a));
  }
  auto need = [&](const char* nm) -> const HostArray* {
    const HostArray* a = m->find(nm);
    if (!a) fail(std::string("nmf_model_create: blob lacks entry ") + nm);
    return a;
  };
  const HostArray *bp = need("body_parent"), *db = need("dof_body"), *at = need("act_type"), *sb = need("seg_body"),
                  *gb = need("geom_body"), *si = need("site_body"), *ns = need("n_sensor"), *os = need("opt_solver"),
                  *star = need("star");
  if (!bp || !db || !at || !sb || !gb || !si || !ns || !os || !star) { delete m; return nullptr; }
  m->nb = (int)bp->count; m->nv = (int)db->count; m->nq = m->nv + 1; m->nu = (int)at->count;
  m->nseg = (int)sb->count; m->ng = (int)gb->count; m->nsite = (int)si->count;
  m->nsensor = ns->i[0]; m->max_iter = os->i[0];
  for (int k = 0; k < 4; ++k) m->star[k] = star->i[(size_t)k];
  return m;
}

extern "C" void nmf_model_destroy(nmf_model* model) { delete model; }

// Most contacts the model's contact set can make in one step: a capsule touches with its two end spheres (and, over a
// terrain with side faces, each of them with one face as well), a hull with up to sem_max_hull_contacts vertices (plus
// one face).  The engine keeps nmf::kMaxCon of them.
extern "C" int nmf_model_contact_bound(const nmf_model* m) {
  if (!m) return fail("nmf_model_contact_bound: null model");
  const HostArray* gt = m->find("geom_type"); const HostArray* so = m->find("sem_options"); const HostArray* tt = m->find("terrain_type");
  if (!gt || !gt->is_int) return fail("nmf_model_contact_bound: model without geom_type");
  const int per_hull = so && so->is_int && so->i.size() > 3 && so->i[3] >= 1 && so->i[3] <= 4 ? so->i[3] : 4;
  const bool faces = tt && tt->is_int && !tt->i.empty() && tt->i[0] != 0 && so && so->i.size() > 4 && so->i[4] != 0;
  int bound = 0;
  for (int t : gt->i) bound += t == nmf::GEOM_CAPSULE ? (faces ? 4 : 2) : per_hull + (faces ? 1 : 0);
  return bound;
}

extern "C" int nmf_batch_set_contact_capacity(nmf_batch* b, int max_contacts) {
  if (!b) return fail("nmf_batch_set_contact_capacity: null batch");
  if (max_contacts < 1) return fail("nmf_batch_set_contact_capacity: max_contacts must be at least 1");
  const int cap = max_contacts > nmf::kMaxCon ? nmf::kMaxCon : max_contacts;
  DEVICE_GUARD(b);      // the caller's current device stays what it was
  b->dm.max_contacts = cap;
  HIP_OK(hipMemcpy(reinterpret_cast<char*>(b->dm_dev) + offsetof(nmf::DevModel, max_contacts), &cap, sizeof(int), hipMemcpyHostToDevice));
  return cap;
}

extern "C" int nmf_model_dims(const nmf_model* m, int32_t out[10]) {
  if (!m) return fail("nmf_model_dims: null model");
  out[0] = m->nq; out[1] = m->nv; out[2] = m->nu; out[3] = m->nb; out[4] = m->nseg; out[5] = m->ng;
  out[6] = m->nsite; out[7] = nmf::kMaxCon; out[8] = 96; out[9] = m->star[0];
  return 0;
}

namespace {

template <class T>
int upload(nmf_batch* b, const std::vector<T>& host, const T** dev) {
  void* p = nullptr;
  size_t bytes = sizeof(T) * (host.empty() ? 1 : host.size());
  HIP_OK(hipMalloc(&p, bytes));
  b->allocs.push_back(p);
  if (!host.empty()) HIP_OK(hipMemcpy(p, host.data(), sizeof(T) * host.size(), hipMemcpyHostToDevice));
  *dev = (const T*)p;
  return 0;
}

// `slot`: address of a DevModel pointer member (the device pass of this file sees those typed as global memory, NMF_G)
int upload_f(nmf_batch* b, const char* name, void* slot) {
  const HostArray* a = b->model->find(name);
  if (!a || a->is_int) return fail(std::string("model lacks float entry ") + name);
  return upload(b, a->f, static_cast<const float**>(slot));
}
int upload_i(nmf_batch* b, const char* name, void* slot) {
  const HostArray* a = b->model->find(name);
  if (!a || !a->is_int) return fail(std::string("model lacks int entry ") + name);
  return upload(b, a->i, static_cast<const int**>(slot));
}

int alloc_field(nmf_batch* b, int field, int width, float** out) {
  void* p = nullptr;
  size_t bytes = sizeof(float) * (size_t)b->n_worlds * (size_t)(width > 0 ? width : 1);
  HIP_OK(hipMalloc(&p, bytes));
  HIP_OK(hipMemset(p, 0, bytes));
  b->allocs.push_back(p);
  b->fields[field] = (float*)p;
  b->widths[field] = width;
  *out = (float*)p;
  return 0;
}

int launch_reset(nmf_batch* b, const uint8_t* mask_dev, hipStream_t stream) {
  DEVICE_GUARD(b);
  dim3 grid((unsigned)b->n_worlds), block(nmf::kWave);
#define NMF_RESET_TOPO(K, TOPO) if (b->topo == K) hipLaunchKernelGGL((nmf::nmf_reset_kernel<TOPO>), grid, block, 0, stream, b->dm_dev, b->st, mask_dev);
#if NMF_HAS_TOPO(0)
  NMF_RESET_TOPO(0, nmf::FlyTopo)
#endif
#if NMF_HAS_TOPO(1)
  NMF_RESET_TOPO(1, nmf::FlyTopoActive)
#endif
#if NMF_HAS_TOPO(2)
  NMF_RESET_TOPO(2, nmf::TreeTopoSmall)
#endif
#if NMF_HAS_TOPO(3)
  NMF_RESET_TOPO(3, nmf::TreeTopo)
#endif
#if NMF_HAS_TOPO(4)
  NMF_RESET_TOPO(4, nmf::FlyTopoBio)
#endif
#if NMF_HAS_TOPO(5)
  NMF_RESET_TOPO(5, nmf::FlyTopoAll)
#endif
#undef NMF_RESET_TOPO
  HIP_OK(hipGetLastError());
  return 0;
}

struct RingArgs { float* ring = nullptr; int stride = 0, every = 0, nj = 0, nact = 0; };

int launch(nmf_batch* b, const nmf::ReplayArgs& rp, int n_steps, hipStream_t stream, const RingArgs& ring = RingArgs()) {
  DEVICE_GUARD(b);      // the caller's current device need not be the batch's, and stays what it was
  b->st.ring = ring.ring; b->st.ring_stride = ring.stride; b->st.obs_every = ring.ring ? ring.every : 0; b->st.ring_nj = ring.nj; b->st.ring_nact = ring.nact;
  // More worlds than resident waves and a launch long enough to cut: chunks whose lengths shrink towards the end of the
  // launch ("guided" sizes: each takes 1 / chunk_div of what is left, at least min_chunk_steps, at most max_chunks chunks)
  // — long items while there is plenty of other work, short ones where they bound the tail.  Measured on 4096 worlds,
  // round 2: whole-launch items 32.0 / 31.9 M env-steps/s (20- / 50-step launches), 7 equal chunks 36.3 / 38.3 M, halving
  // chunks (chunk_div 2: 10 + 5 + 3 + 1 + 1 steps) 42.4 / 44.4 M; a last chunk of one step (min_chunk_steps 1) is worth
  // +1.2 % on 50-step launches.  Round 4 (a step a quarter cheaper, the hand-over the same): fewer, longer chunks — chunk_div
  // 1.4 / 1.5 / 1.6 / 1.7 / 1.8 / 2.0: 51.4 / 53.4 / 53.6 / 53.4 / 53.3 / 52.5 M on 20-step launches, 53.4 / 54.5 / 55.3 / 55.5 / 55.1 / 55.2 M on 50-step ones.
  const bool oversub = b->n_worlds > b->resident_waves;
  int n_chunks = 1;
  b->st.n_chunks = 1; b->st.csched = b->csched_buf; b->st.handoff = b->handoff_buf; b->st.handoff_stride = b->handoff_stride;
  if (oversub && b->chunking && b->csched_buf && b->handoff_buf && n_steps >= 2 * b->min_chunk_steps) {
    int start = 0, c = 0;
    while (start < n_steps && c < b->max_chunks) {
      // (the library's own plan — no chunk_div option given: launches of more than 64 steps halve; on flat ground with a leg-chain
      // skeleton launches of up to 30 steps take 1.6 instead of 1.7 — 20 steps = 13 + 5 + 2, three chunks instead of 12 + 5 + 2 + 1:
      // 60.05 -> 60.6 M at the driver's arguments, round 6; 30 steps tie, 40 and 50 steps keep 1.7: 63.0 against 62.5 M)
      const double div = !b->chunk_div_short ? b->chunk_div : n_steps > 64 ? 2.0 : (b->chunk_div < 1.75 && n_steps <= 30 ? 1.6 : b->chunk_div);
      int len = (int)std::ceil((n_steps - start) / div);
      len = std::max(len, b->min_chunk_steps);
      if (c == b->max_chunks - 1 || n_steps - start - len < b->min_chunk_steps) len = n_steps - start;
      b->st.chunk_start[c++] = start;
      start += len;
    }
    b->st.chunk_start[c] = n_steps;
    n_chunks = c;
    b->st.n_chunks = c;
  }
  b->st.sched_mode = n_chunks > 1 ? 1 : 0;
  // chunked: one persistent workgroup per resident wave pulls (chunk, world) items; plain: one workgroup per world
  dim3 grid(n_chunks > 1 ? (unsigned)std::min(b->resident_waves, b->n_worlds * n_chunks) : (unsigned)b->n_worlds), block(nmf::kWave);
  // more worlds than resident waves: the launch runs in rounds; the measured policy picks the world order (nmf_order_kernel)
  b->st.order = nullptr; b->st.sched = nullptr;
  if (oversub && b->order_buf && b->sched_buf && b->order_policy != 2) {
    const int policy = b->order_policy == 3 ? (n_steps <= 64 ? 1 : -1) : b->order_policy;
    if (!(policy == 1 && b->order_valid && ++b->order_age < b->order_every)) {
      hipLaunchKernelGGL(nmf::nmf_order_kernel, dim3(1), dim3(1024), 0, stream, b->st.cost, b->n_worlds, b->order_buf, b->sched_buf, n_steps, policy);
      b->order_age = 0; b->order_valid = true;
    }
    b->st.order = b->order_buf;
    if (policy < 0) b->st.sched = b->sched_buf;
  }
  const bool weld = b->dm.weld_active != 0, terrain = b->dm.terrain_type != 0;
#define NMF_LAUNCH(TOPO, WELD) hipLaunchKernelGGL((nmf::nmf_step_kernel<TOPO, WELD>), grid, block, b->lds_pad, stream, b->dm_dev, b->st, rp, n_steps)
#define NMF_LAUNCH_TOPO(K, TOPO) if (b->topo == K) { if (weld) NMF_LAUNCH(TOPO, true); else if (terrain) NMF_LAUNCH(nmf::Terrain<TOPO>, false); else NMF_LAUNCH(TOPO, false); }
#if NMF_HAS_TOPO(0)
  NMF_LAUNCH_TOPO(0, nmf::FlyTopo)
#endif
#if NMF_HAS_TOPO(1)
  NMF_LAUNCH_TOPO(1, nmf::FlyTopoActive)
#endif
#if NMF_HAS_TOPO(2)
  NMF_LAUNCH_TOPO(2, nmf::TreeTopoSmall)
#endif
#if NMF_HAS_TOPO(3)
  NMF_LAUNCH_TOPO(3, nmf::TreeTopo)
#endif
#if NMF_HAS_TOPO(4)
  NMF_LAUNCH_TOPO(4, nmf::FlyTopoBio)
#endif
#if NMF_HAS_TOPO(5)
  NMF_LAUNCH_TOPO(5, nmf::FlyTopoAll)
#endif
#undef NMF_LAUNCH_TOPO
#undef NMF_LAUNCH
  HIP_OK(hipGetLastError());
  return 0;
}

}  // namespace

namespace {
template <class T> struct TopoTag { using type = T; };
// Development overrides from the process environment — honoured only under NMF_ALLOW_ENV=1 (the one getenv gate of this
// library): a stray NMF_* variable in a user's shell never changes what a batch runs.
const char* dev_env(const char* name) {
  static const bool allowed = [] { const char* e = getenv("NMF_ALLOW_ENV"); return e && atoi(e) != 0; }();
  return allowed ? getenv(name) : nullptr;
}
}  // namespace

extern "C" nmf_batch* nmf_batch_create(const nmf_model* model, int n_worlds, int device) {
  return nmf_batch_create_ex(model, n_worlds, device, nullptr);
}

extern "C" nmf_batch* nmf_batch_create_ex(const nmf_model* model, int n_worlds, int device, const nmf_batch_options* options) {
  g_err.clear();
  nmf_batch_options opt{};
  if (options) {
    if (options->struct_size < (int32_t)sizeof(int32_t) || options->struct_size > (int32_t)sizeof(nmf_batch_options)) { fail("nmf_batch_create_ex: options->struct_size does not match this library"); return nullptr; }
    memcpy(&opt, options, (size_t)options->struct_size);
  }
  if (!model) { fail("nmf_batch_create: null model"); return nullptr; }
  if (n_worlds <= 0) { fail("nmf_batch_create: n_worlds must be positive"); return nullptr; }
  int topo = -1;
  if (model->star[0] == 1 && model->star[1] == 6 && model->star[2] == 11 && model->star[3] == 8) topo = 0;
  if (model->star[0] == 1 && model->star[1] == 6 && model->star[2] == 7 && model->star[3] == 4) topo = 1;
  if (topo >= 0) {  // the chain-star kernels hard-wire the per-leg hinge layout (and 48 controls); anything else takes the tree kernel
    const HostArray* dn = model->find("body_dofnum");
    const int pat0[8] = {3, 2, 1, 1, 1, 1, 1, 1}, pat1[4] = {3, 2, 1, 1};
    const int* pat = topo == 0 ? pat0 : pat1;
    const int nbl = topo == 0 ? 8 : 4;
    bool ok = dn && dn->is_int && (int)dn->i.size() == 1 + 6 * nbl && dn->i[0] == 6;
    for (int b = 1; ok && b < 1 + 6 * nbl; ++b) ok = dn->i[(size_t)b] == pat[(b - 1) % nbl];
    if (!ok || model->nu > nmf::kMaxCtrl) topo = -1;
  }
  // the full-body skeletons (ALL_BIOLOGICAL, ALL_POSSIBLE): six identical leg chains at the END of the body order, the
  // rest of the body (20 bodies, 60 dofs) between the root and the legs -> hybrid kernels (legs unrolled, rest as a tree)
  std::vector<char> in_tree;          // bodies the tree tables cover (hybrid: root + rest; tree kernels: all)
  if (topo < 0) {
    const HostArray* bp = model->find("body_parent");
    const HostArray* dn = model->find("body_dofnum");
    const int patB[8] = {3, 2, 1, 1, 1, 1, 1, 1}, patA[8] = {3, 3, 3, 3, 3, 3, 3, 3};
    for (int cand = 4; cand <= 5 && topo < 0 && bp && dn; ++cand) {
      const int* pat = cand == 4 ? patB : patA;
      const int nb = model->nb, lb0 = 21, want_nv = cand == 4 ? 132 : 210;
      bool ok = nb == 69 && model->nv == want_nv && (int)dn->i.size() == nb && dn->i[0] == 6 && model->nu <= (cand == 4 ? nmf::FlyTopoBio::kCtrl : nmf::FlyTopoAll::kCtrl);
      int rest_v = 0;
      for (int bb = 1; ok && bb < lb0; ++bb) { rest_v += dn->i[(size_t)bb]; ok = bp->i[(size_t)bb] >= 0 && bp->i[(size_t)bb] < lb0 && bp->i[(size_t)bb] < bb; }
      ok = ok && rest_v == 60;
      for (int bb = lb0; ok && bb < nb; ++bb) {
        const int l = (bb - lb0) % 8;
        ok = dn->i[(size_t)bb] == pat[l] && bp->i[(size_t)bb] == (l == 0 ? 0 : bb - 1);
      }
      if (ok) { topo = cand; in_tree.assign((size_t)nb, 0); for (int bb = 0; bb < lb0; ++bb) in_tree[(size_t)bb] = 1; }
    }
  }
  // anything else (custom skeletons): the general-tree kernel, up to 72 bodies / 216 dofs
  std::vector<int> tree_body, child_start, child_count, lvl_start;
  if (topo < 0 || topo >= 4) {
    if (topo < 0) topo = model->nv <= nmf::TreeTopoSmall::NV && model->nu <= nmf::TreeTopoSmall::kCtrl ? 2 : 3;
    const HostArray* bp = model->find("body_parent");
    const HostArray* dn = model->find("body_dofnum");
    const HostArray* gb = model->find("geom_body");
    if (!bp || !dn || !gb || model->nb > nmf::TreeTopo::NB || model->nv > nmf::TreeTopo::NV || dn->i.empty() || dn->i[0] != 6) {
      fail("nmf_batch_create: the general-tree kernel takes a free-floating root and up to 72 bodies / 216 dofs"); return nullptr;
    }
    const int nb = model->nb;
    if (in_tree.empty()) in_tree.assign((size_t)nb, 1);
    for (int bb = 1; bb < nb; ++bb)
      if (bp->i[(size_t)bb] < 0 || bp->i[(size_t)bb] >= bb) { fail("nmf_batch_create: bodies must be ordered parents first"); return nullptr; }
    for (size_t g = 1; g < gb->i.size(); ++g)
      if (gb->i[g] < gb->i[g - 1]) { fail("nmf_batch_create: contact geoms must be ordered by body"); return nullptr; }
    // breadth-first order: level by level, the children of a body contiguous
    std::vector<int> depth((size_t)nb, 0);
    int maxd = 0;
    for (int bb = 1; bb < nb; ++bb) {
      depth[(size_t)bb] = depth[(size_t)bp->i[(size_t)bb]] + 1;
      if (in_tree[(size_t)bb]) maxd = std::max(maxd, depth[(size_t)bb]);
    }
    if (maxd + 2 > 18) { fail("nmf_batch_create: kinematic tree deeper than 16 levels"); return nullptr; }
    tree_body.push_back(0); lvl_start.push_back(0);
    child_start.assign((size_t)nb, 0); child_count.assign((size_t)nb, 0);
    for (int lv = 0; lv <= maxd; ++lv) {
      const int k0 = lvl_start[(size_t)lv], k1 = (int)tree_body.size();
      lvl_start.push_back(k1);
      for (int k = k0; k < k1; ++k) {
        const int par = tree_body[(size_t)k];
        child_start[(size_t)par] = (int)tree_body.size();
        for (int bb = 1; bb < nb; ++bb) if (bp->i[(size_t)bb] == par && in_tree[(size_t)bb]) { tree_body.push_back(bb); child_count[(size_t)par]++; }
      }
      if (k1 - k0 > nmf::kWave) { fail("nmf_batch_create: more than 64 bodies on one tree level"); return nullptr; }
    }
    // lvl_start has maxd + 2 entries: starts of levels 0..maxd and the end
  }
  if (model->ng > 2 * nmf::kWave) { fail("nmf_batch_create: more than 128 contact geoms"); return nullptr; }
  if (model->nu > (topo >= 2 ? nmf::TreeTopo::kCtrl : nmf::kMaxCtrl)) { fail("nmf_batch_create: too many actuators (48 for the leg skeletons, 224 otherwise)"); return nullptr; }
  DeviceGuard guard(device);            // allocations and the first reset run on `device`; the caller's device comes back on return
  if (guard.err != hipSuccess) { fail("nmf_batch_create: hipSetDevice failed (no MI355X visible?)"); return nullptr; }
  auto* b = new nmf_batch();
  b->model = model; b->n_worlds = n_worlds; b->device = device; b->topo = topo;
  nmf::DevModel& d = b->dm;
  d.nb = model->nb; d.nv = model->nv; d.nq = model->nq; d.nu = model->nu; d.ng = model->ng;
  d.nseg = model->nseg; d.nsite = model->nsite; d.nsensor = model->nsensor; d.max_iter = model->max_iter;
  auto scalar = [&](const char* nm, int k) { const HostArray* a = model->find(nm); return a && !a->is_int && (int)a->f.size() > k ? a->f[(size_t)k] : 0.f; };
  d.timestep = scalar("opt_timestep", 0); d.tolerance = scalar("opt_tolerance", 0);
  d.hull_skin = scalar("hull_skin", 0); d.meaninertia = scalar("stat_meaninertia", 0);
  for (int k = 0; k < 3; ++k) d.gravity[k] = scalar("opt_gravity", k);
  for (int k = 0; k < 4; ++k) d.plane[k] = scalar("plane", k);
  for (int k = 0; k < 5; ++k) d.terrain[k] = scalar("terrain_params", k);
  { const HostArray* wa = model->find("weld_active"); d.weld_active = wa && wa->is_int && !wa->i.empty() ? wa->i[0] : 0;
    for (int k = 0; k < 3; ++k) d.weld_pos[k] = scalar("weld_params", k);
    for (int k = 0; k < 4; ++k) d.weld_quat[k] = scalar("weld_params", 3 + k);
    for (int k = 0; k < 2; ++k) d.weld_solref[k] = scalar("weld_params", 7 + k);
    for (int k = 0; k < 5; ++k) d.weld_solimp[k] = scalar("weld_params", 9 + k);
    for (int k = 0; k < 2; ++k) d.weld_invweight[k] = scalar("weld_params", 14 + k); }
  { const HostArray* so = model->find("sem_options");
    if (!so || !so->is_int || so->i.size() < 5) { delete b; fail("nmf_batch_create: model lacks sem_options"); return nullptr; }
    d.sem_terrain_walls = so->i[4];
    d.sem_pyramid_plain = so->i[0]; d.sem_adhesion_fused = so->i[1]; d.sem_sensor_contact_frame = so->i[2];
    d.sem_max_hull_contacts = so->i[3] >= 1 && so->i[3] <= 4 ? so->i[3] : 4; }
  { const HostArray* tt = model->find("terrain_type"); d.terrain_type = tt && tt->is_int && !tt->i.empty() ? tt->i[0] : 0; }
  int rc = 0;
#define UF(n) rc |= upload_f(b, #n, (void*)&d.n)
#define UI(n) rc |= upload_i(b, #n, (void*)&d.n)
  UF(body_pos); UF(body_quat); UF(body_mass); UF(body_ipos); UF(body_inertia);
  UI(body_dofadr); UI(body_dofnum); UI(dof_body);
  UF(dof_axis); UF(dof_armature); UF(dof_damping); UF(dof_stiffness); UF(dof_springref);
  UI(seg_body); UF(seg_pos); UF(seg_quat); UI(site_body); UF(site_pos);
  UI(act_type); UI(act_trn); UI(act_limited); UI(act_geom); UF(act_gain); UF(act_bias); UF(act_forcerange); UF(act_ctrlrange);
  UF(key_qpos); UF(key_ctrl);
  d.act_general = nullptr;
  if (const HostArray* ag = model->find("act_general")) {        // optional: models with intvelocity / damper / cylinder / muscle actuators or shared dofs
    if (ag->is_int || (int)ag->f.size() != model->nu * nmf::kActGen) { rc |= fail("nmf_batch_create: act_general must be float [nu][32]"); }
    else {
      UF(act_general);
      // the affine pass of the stepping kernel writes every actuator's force to its dof with a plain store (one lane per actuator,
      // one actuator per dof): the general actuators' zero force goes to dof 0 — no affine actuator drives the root — instead of
      // racing with the affine actuator that may share their dof; the general pass reads the dof from the row's flags
      std::vector<int> trn = model->find("act_trn")->i;
      for (int u = 0; u < model->nu; ++u) if ((int)ag->f[(size_t)u * nmf::kActGen] & 1) trn[(size_t)u] = 0;
      rc |= upload(b, trn, reinterpret_cast<const int**>((void*)&d.act_trn));
    }
  }
  UI(geom_body); UI(geom_type); UI(geom_hulladr); UI(geom_hullnum); UI(geom_sensor);
  UF(geom_p0); UF(geom_p1); UF(geom_radius); UF(geom_bsphere); UF(geom_invweight0); UF(hull_vert);
  UF(pair_friction); UF(pair_solref); UF(pair_solimp); UF(pair_margin);
#undef UF
#undef UI
  if (topo >= 2) {
    const HostArray* bp = model->find("body_parent");
    rc |= upload(b, bp->i, reinterpret_cast<const int**>((void*)&d.body_parent));
    rc |= upload(b, tree_body, reinterpret_cast<const int**>((void*)&d.tree_body));
    rc |= upload(b, child_start, reinterpret_cast<const int**>((void*)&d.tree_child_start));
    rc |= upload(b, child_count, reinterpret_cast<const int**>((void*)&d.tree_child_count));
    d.tree_nlevel = (int)lvl_start.size() - 1;
    for (size_t k = 0; k < 18; ++k) d.tree_lvl_start[k] = k < lvl_start.size() ? lvl_start[k] : (int)tree_body.size();
    d.rest_fast = 0; d.rest_pack = nullptr;
    if (topo >= 4) {
      const HostArray* dn = model->find("body_dofnum");
      const HostArray* da = model->find("body_dofadr");
      const int nl = (int)lvl_start.size() - 2;                 // levels below the root
      // NMF_DISABLE_REST_FAST: diagnostic switch, forces the table-driven level passes (tests cover both paths)
      bool fast = da && nl <= nmf::kRestLevels && !opt.rest_slow && !dev_env("NMF_DISABLE_REST_FAST");
      std::vector<int> pack((size_t)nmf::kRestLevels * 16, -1);
      for (int lv = 1; fast && lv <= nl; ++lv) {
        const int k0 = lvl_start[(size_t)lv], k1 = lvl_start[(size_t)lv + 1];
        fast = k1 - k0 <= 8;
        for (int k = k0; fast && k < k1; ++k) {
          const int bb = tree_body[(size_t)k];
          fast = dn->i[(size_t)bb] == 3 && da->i[(size_t)bb] < 256 && child_count[(size_t)bb] < 256 && child_start[(size_t)bb] < 256;
          pack[(size_t)((lv - 1) * 8 + (k - k0)) * 2] = bb | (bp->i[(size_t)bb] << 8) | (da->i[(size_t)bb] << 16) | (child_count[(size_t)bb] << 24);
          pack[(size_t)((lv - 1) * 8 + (k - k0)) * 2 + 1] = child_start[(size_t)bb] | (k << 8);
        }
      }
      if (fast) { d.rest_fast = 1; const int* pp = nullptr; rc |= upload(b, pack, &pp); *reinterpret_cast<const int**>((void*)&d.rest_pack) = pp; }
    }
  } else {
    d.body_parent = d.tree_body = d.tree_child_start = d.tree_child_count = nullptr; d.tree_nlevel = 0; d.rest_fast = 0; d.rest_pack = nullptr;
  }
  d.noslip_iter = 0;
  if (const HostArray* os2 = model->find("opt_solver")) { if (os2->i.size() > 1) d.noslip_iter = os2->i[1]; }
  d.max_contacts = nmf::kMaxCon;
  d.solver_flags = opt.solver & 7;
  if (const char* e = dev_env("NMF_SOLVER")) {      // primal | nohist | nofallback (default: contact-space solve with the active-set history)
    const std::string v(e);
    d.solver_flags = v == "primal" ? 1 : v == "nohist" ? 2 : v == "nofallback" ? 4 : 0;
  }
  if (rc == 0) {
    void* p = nullptr;
    if (hipMalloc(&p, sizeof(nmf::DevModel)) != hipSuccess ||
        hipMemcpy(p, &d, sizeof(nmf::DevModel), hipMemcpyHostToDevice) != hipSuccess) { fail("nmf_batch_create: model upload failed"); rc = -1; }
    else { b->allocs.push_back(p); b->dm_dev = (nmf::DevModel*)p; }
  }
  nmf::DevState& st = b->st;
  st.n_worlds = n_worlds;
  rc |= alloc_field(b, NMF_QPOS, model->nq, &st.qpos);
  rc |= alloc_field(b, NMF_QVEL, model->nv, &st.qvel);
  rc |= alloc_field(b, NMF_CTRL, model->nu, &st.ctrl);
  rc |= alloc_field(b, NMF_QACC_WARMSTART, model->nv, &st.qacc_ws);
  rc |= alloc_field(b, NMF_SEG_XPOS, model->nseg * 3, &st.seg_xpos);
  rc |= alloc_field(b, NMF_SEG_XQUAT, model->nseg * 4, &st.seg_xquat);
  rc |= alloc_field(b, NMF_SITE_XPOS, model->nsite * 3, &st.site_xpos);
  rc |= alloc_field(b, NMF_ACTUATOR_FORCE, model->nu, &st.actuator_force);
  rc |= alloc_field(b, NMF_SENSORDATA, 96, &st.sensordata);
  rc |= alloc_field(b, NMF_TIME, 1, &st.time);
  rc |= alloc_field(b, NMF_STATS, 8, &st.stats);
  rc |= alloc_field(b, NMF_QACC, model->nv, &st.qacc);
  rc |= alloc_field(b, NMF_COST, 1, &st.cost);
  { float* p = nullptr; rc |= alloc_field(b, NMF_STATS_SUM, 16, &p); st.stats_sum = reinterpret_cast<unsigned int*>(p); }   // uint32 counters
  rc |= alloc_field(b, NMF_CONTACT_GEOM, nmf::kMaxCon, &st.contact_geom);
  rc |= alloc_field(b, NMF_ACT, model->nu, &st.act);
  {
    void* p = nullptr;
    if (hipMalloc(&p, sizeof(int) * (size_t)n_worlds) == hipSuccess) { b->allocs.push_back(p); b->order_buf = (int*)p; }
    else rc |= fail("nmf_batch_create: out of device memory");
    p = nullptr;
    if (hipMalloc(&p, sizeof(nmf::ChunkSched)) == hipSuccess) {
      (void)hipMemset(p, 0, sizeof(nmf::ChunkSched));
      b->allocs.push_back(p); b->csched_buf = (nmf::ChunkSched*)p;
    } else rc |= fail("nmf_batch_create: out of device memory");
    p = nullptr;
    b->handoff_stride = (model->nq + 2 * model->nv + model->nu + 6 + nmf::kActHistWords + 63) / 64 * 64;      // state, controls, clock + 5 running sums, active-set history
    if (hipMalloc(&p, sizeof(unsigned long long) * (size_t)n_worlds * (size_t)b->handoff_stride) == hipSuccess) {
      (void)hipMemset(p, 0, sizeof(unsigned long long) * (size_t)n_worlds * (size_t)b->handoff_stride);   // tag 0 = no launch's
      b->allocs.push_back(p); b->handoff_buf = (unsigned long long*)p;
    } else rc |= fail("nmf_batch_create: out of device memory");
    p = nullptr;
    if (hipMalloc(&p, sizeof(unsigned int) * (size_t)n_worlds * nmf::kActHistWords) == hipSuccess) {
      (void)hipMemset(p, 0, sizeof(unsigned int) * (size_t)n_worlds * nmf::kActHistWords);
      b->allocs.push_back(p); st.act_hist = (unsigned int*)p;
    } else rc |= fail("nmf_batch_create: out of device memory");
    st.dual_scratch = nullptr;
    if (topo == 5) {       // ALL_POSSIBLE: the contact-space solve's leg factors live in HBM, one block per workgroup of a launch (<= n_worlds)
      p = nullptr;
      if (hipMalloc(&p, sizeof(float) * (size_t)n_worlds * nmf::kDualScratchFloats) == hipSuccess) { b->allocs.push_back(p); st.dual_scratch = (float*)p; }
      else rc |= fail("nmf_batch_create: out of device memory");
    }
    st.noslip_buf = nullptr;
    if (d.noslip_iter > 0) {       // CPU flavour: scratch of the primal path's noslip pass (157 KB per world)
      p = nullptr;
      if (hipMalloc(&p, sizeof(float) * (size_t)n_worlds * nmf::kNoslipFloats) == hipSuccess) {
        (void)hipMemset(p, 0, sizeof(float) * (size_t)n_worlds * nmf::kNoslipFloats);
        b->allocs.push_back(p); st.noslip_buf = (float*)p;
      } else rc |= fail("nmf_batch_create: out of device memory");
    }
    p = nullptr;
    if (hipMalloc(&p, 2 * sizeof(unsigned long long)) == hipSuccess) {
      (void)hipMemset(p, 0, 2 * sizeof(unsigned long long));
      b->allocs.push_back(p); b->clock_probe_buf = (unsigned long long*)p;
    } else rc |= fail("nmf_batch_create: out of device memory");
    st.clock_probe = b->clock_probe_buf;
    b->chunking = opt.sched != 1;
    if (opt.order) b->order_policy = opt.order == 1 ? 0 : opt.order == 2 ? 1 : opt.order == 3 ? 2 : opt.order == 4 ? -1 : 3;
    if (opt.order_every > 0) b->order_every = opt.order_every;
    if (opt.max_chunks > 0) b->max_chunks = std::max(1, std::min(16, opt.max_chunks));
    // (flat ground, leg-chain skeleton: a world's cost varies least and a step is cheapest against the hand-over — fewer, longer
    // chunks; terrains and the full-body skeletons keep the halving plan: blocks 34.2 vs 32.4 M, ALL_BIOLOGICAL 30.7 vs 30.2 M)
    // (and launches of more than 64 steps: 250-step launches 56.1 M halving, 54.5 M with 1.6)
    // (round 5, one contact-space solve for every walking step: 1.5 / 1.6 / 1.7 / 1.8 / 2.0 = 56.7 / 56.6 / 56.7 / 56.6 / 55.8 M on 20-step
    // launches, 58.7 / 58.9 / 59.2 / 59.0 / 58.9 M on 50-step ones)
    b->chunk_div = (b->dm.terrain_type == 0 && topo < 2) ? 1.7 : 2.0;
    b->chunk_div_short = true;
    if (opt.chunk_div > 1.f) { b->chunk_div = opt.chunk_div; b->chunk_div_short = false; }
    if (opt.min_chunk_steps > 0) b->min_chunk_steps = opt.min_chunk_steps;
    if (const char* e = dev_env("NMF_SCHED")) b->chunking = std::string(e) != "plain";
    if (const char* e = dev_env("NMF_ORDER")) {
      const std::string v(e);
      b->order_policy = v == "inorder" ? 0 : v == "costliest" ? 1 : v == "none" ? 2 : v == "policy" ? -1 : 3;
    }
    if (const char* e = dev_env("NMF_ORDER_EVERY")) b->order_every = std::max(1, atoi(e));
    if (const char* e = dev_env("NMF_MAX_CHUNKS")) b->max_chunks = std::max(1, std::min(16, atoi(e)));
    if (const char* e = dev_env("NMF_CHUNK_DIV")) { b->chunk_div = std::max(1.0, atof(e)); b->chunk_div_short = false; }
    if (const char* e = dev_env("NMF_MIN_CHUNK_STEPS")) b->min_chunk_steps = std::max(1, atoi(e));
    // (activations live in HBM and are advanced in place every step by the lane that owns the actuator: a world's steps must stay
    // on one workgroup within a launch — whole-launch work items for the models that have general actuators)
    if (d.act_general) b->chunking = false;
    p = nullptr;
    if (hipMalloc(&p, sizeof(nmf::SchedState)) == hipSuccess) {
      (void)hipMemset(p, 0, sizeof(nmf::SchedState));
      b->allocs.push_back(p); b->sched_buf = (nmf::SchedState*)p;
    } else rc |= fail("nmf_batch_create: out of device memory");
    st.sched = nullptr;
    st.order = nullptr;
    hipDeviceProp_t prop;
    // flies (= single-wave workgroups) a CU holds at once: asked of the runtime for the kernel this batch will launch
    // (LDS- or register-limited, whichever binds); fallback = the LDS-limited figures of the shipped build
    int per_cu = topo < 2 ? 8 : (topo == 2 ? 4 : (topo == 3 ? 3 : (topo == 4 ? 8 : 5)));
    {
      const bool weld = b->dm.weld_active != 0, terrain = b->dm.terrain_type != 0;
      if (weld && terrain) { nmf_batch_destroy(b); fail("nmf_batch_create: a tethered world has no terrain"); return nullptr; }
      const void* fn = nullptr;
#define NMF_FN(K, TOPO) if (topo == K) fn = weld ? reinterpret_cast<const void*>(&nmf::nmf_step_kernel<TOPO, true>) : terrain ? reinterpret_cast<const void*>(&nmf::nmf_step_kernel<nmf::Terrain<TOPO>, false>) : reinterpret_cast<const void*>(&nmf::nmf_step_kernel<TOPO, false>);
#if NMF_HAS_TOPO(0)
      NMF_FN(0, nmf::FlyTopo)
#endif
#if NMF_HAS_TOPO(1)
      NMF_FN(1, nmf::FlyTopoActive)
#endif
#if NMF_HAS_TOPO(2)
      NMF_FN(2, nmf::TreeTopoSmall)
#endif
#if NMF_HAS_TOPO(3)
      NMF_FN(3, nmf::TreeTopo)
#endif
#if NMF_HAS_TOPO(4)
      NMF_FN(4, nmf::FlyTopoBio)
#endif
#if NMF_HAS_TOPO(5)
      NMF_FN(5, nmf::FlyTopoAll)
#endif
      if (!fn) { nmf_batch_destroy(b); fail("nmf_batch_create: this build of the library has no kernel for the model's skeleton (NMF_TOPO_MASK)"); return nullptr; }
#undef NMF_FN
      int nblk = 0;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nblk, fn, nmf::kWave, 0) == hipSuccess && nblk > 0) per_cu = nblk;
      b->step_fn = fn;
      // options.flies_per_cu below the kernel's own residency: idle LDS per workgroup so that k workgroups fit a CU and k + 1 do
      // not — as little of it as that takes (allocation granule 512 B), so the CU keeps LDS for another stream's kernel
      int want = opt.flies_per_cu;
      if (const char* e = dev_env("NMF_FLIES_PER_CU")) want = atoi(e);
      hipFuncAttributes fa;
      if (want > 0 && want < per_cu && hipFuncGetAttributes(&fa, fn) == hipSuccess) {
        int lds_cu = 0;
        if (hipDeviceGetAttribute(&lds_cu, hipDeviceAttributeMaxSharedMemoryPerMultiprocessor, device) != hipSuccess || lds_cu <= 0) lds_cu = 160 * 1024;
        const int granule = 512;
        int per_wg = (lds_cu / (want + 1) / granule + 1) * granule;            // the smallest allocation of which want + 1 do not fit
        if (per_wg * want <= lds_cu && per_wg > (int)fa.sharedSizeBytes) {
          b->lds_pad = (unsigned)(per_wg - (int)fa.sharedSizeBytes);
          int nblk2 = 0;
          if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nblk2, fn, nmf::kWave, b->lds_pad) == hipSuccess && nblk2 > 0) per_cu = std::min(per_cu, nblk2);
          else per_cu = want;
        }
      }
      if (const char* e = dev_env("NMF_LDS_PAD")) b->lds_pad = (unsigned)atoi(e);
      b->per_cu = per_cu;
    }
    b->resident_waves = (hipGetDeviceProperties(&prop, device) == hipSuccess ? prop.multiProcessorCount : 256) * per_cu;
  }
  if (rc != 0 || nmf_reset(b, nullptr) != 0 || hipDeviceSynchronize() != hipSuccess) {
    std::string keep = g_err.empty() ? std::string("nmf_batch_create: device initialisation failed") : g_err;
    nmf_batch_destroy(b);
    g_err = keep;
    return nullptr;
  }
  return b;
}

extern "C" void nmf_batch_destroy(nmf_batch* b) {
  if (!b) return;
  DeviceGuard guard(b->device);
  for (void* p : b->allocs) (void)hipFree(p);
  for (nmf_eye_plan* q : b->eye_plans) nmf_eye_plan_destroy(q);
  delete b;
}

extern "C" int nmf_batch_n_worlds(const nmf_batch* b) { return b ? b->n_worlds : 0; }

extern "C" int nmf_batch_info(const nmf_batch* b, int32_t out[16]) {
  if (!b || !out) return fail("nmf_batch_info: null argument");
  const bool weld = b->dm.weld_active != 0, terrain = b->dm.terrain_type != 0;
  int flavour = 0, maxcon = 0;
  auto dual_of = [&](auto topo_tag) {
    using TP = typename decltype(topo_tag)::type;
    if (!weld && !(b->dm.solver_flags & 1)) { flavour = nmf::kDualS<TP> ? 1 : nmf::kDualH<TP> ? 2 : 0; maxcon = flavour ? nmf::kDualMaxCon<TP> : 0; }
  };
  switch (b->topo) {
#if NMF_HAS_TOPO(0)
    case 0: dual_of(TopoTag<nmf::FlyTopo>{}); break;
#endif
#if NMF_HAS_TOPO(1)
    case 1: dual_of(TopoTag<nmf::FlyTopoActive>{}); break;
#endif
#if NMF_HAS_TOPO(4)
    case 4: dual_of(TopoTag<nmf::FlyTopoBio>{}); break;
#endif
#if NMF_HAS_TOPO(5)
    case 5: dual_of(TopoTag<nmf::FlyTopoAll>{}); break;
#endif
    default: break;
  }
  out[0] = b->topo; out[1] = terrain ? 1 : 0; out[2] = weld ? 1 : 0; out[3] = flavour; out[4] = maxcon;
  out[5] = b->per_cu; out[6] = b->resident_waves; out[7] = b->chunking ? 1 : 0; out[8] = b->max_chunks;
  out[9] = (int32_t)std::lround(1000.0 * b->chunk_div); out[10] = b->order_policy; out[11] = b->dm.solver_flags;
  out[12] = b->dm.noslip_iter; out[13] = b->dm.max_contacts; out[14] = 0; out[15] = 0;
  hipFuncAttributes fa;
  if (b->step_fn && hipFuncGetAttributes(&fa, b->step_fn) == hipSuccess) { out[14] = (int32_t)fa.sharedSizeBytes; out[15] = fa.numRegs; }
  return 0;
}

extern "C" int nmf_reset(nmf_batch* b, void* stream) {
  if (!b) return fail("nmf_reset: null batch");
  b->steps = 0;
  return launch_reset(b, nullptr, (hipStream_t)stream);
}

extern "C" int nmf_reset_worlds(nmf_batch* b, const uint8_t* mask_dev, void* stream) {
  if (!b) return fail("nmf_reset_worlds: null batch");
  if (!mask_dev) return fail("nmf_reset_worlds: null mask");
  return launch_reset(b, mask_dev, (hipStream_t)stream);
}

extern "C" int nmf_step(nmf_batch* b, int n_steps, void* stream) {
  if (!b) return fail("nmf_step: null batch");
  if (n_steps <= 0) return fail("nmf_step: n_steps must be positive");
  nmf::ReplayArgs rp{nullptr, nullptr, 1, 0, 0};
  b->steps += n_steps;
  return launch(b, rp, n_steps, (hipStream_t)stream);
}

extern "C" int nmf_step_replay(nmf_batch* b, const float* table_dev, int table_steps, int n_act,
                               const int32_t* act_ids_dev, int start, int n_steps, void* stream) {
  if (!b) return fail("nmf_step_replay: null batch");
  if (n_steps <= 0) return fail("nmf_step_replay: n_steps must be positive");
  if (!table_dev || !act_ids_dev || table_steps <= 0 || n_act <= 0 || n_act > b->model->nu)
    return fail("nmf_step_replay: bad replay table arguments");
  nmf::ReplayArgs rp{table_dev, act_ids_dev, table_steps, n_act, ((start % table_steps) + table_steps) % table_steps};
  b->steps += n_steps;
  return launch(b, rp, n_steps, (hipStream_t)stream);
}

extern "C" int nmf_step_record(nmf_batch* b, const float* table_dev, int table_steps, int n_act_table, const int32_t* act_ids_dev, int start,
                               int n_steps, int obs_every, int n_joint, int n_act, float* ring_dev, int row_stride, void* stream) {
  if (!b) return fail("nmf_step_record: null batch");
  if (n_steps <= 0 || obs_every <= 0) return fail("nmf_step_record: n_steps and obs_every must be positive");
  if (n_steps % obs_every) return fail("nmf_step_record: n_steps must be a multiple of obs_every (the trailing steps would not be recorded)");
  if (!ring_dev) return fail("nmf_step_record: null ring");
  const nmf_model* m = b->model;
  if (n_joint < 0 || n_joint > m->nv - 6 || n_act < 0 || n_act > m->nu || row_stride < 2 * n_joint + n_act + 96)
    return fail("nmf_step_record: need 0 <= n_joint <= nv - 6, 0 <= n_act <= nu and row_stride >= 2 n_joint + n_act + 96");
  nmf::ReplayArgs rp{nullptr, nullptr, 1, 0, 0};
  if (table_dev) {
    if (!act_ids_dev || table_steps <= 0 || n_act_table <= 0 || n_act_table > m->nu) return fail("nmf_step_record: bad replay table arguments");
    rp = nmf::ReplayArgs{table_dev, act_ids_dev, table_steps, n_act_table, ((start % table_steps) + table_steps) % table_steps};
  }
  RingArgs ring;
  ring.ring = ring_dev; ring.stride = row_stride; ring.every = obs_every; ring.nj = n_joint; ring.nact = n_act;
  b->steps += n_steps;
  return launch(b, rp, n_steps, (hipStream_t)stream, ring);
}

extern "C" float* nmf_field_ptr(nmf_batch* b, int field, int32_t* width) {
  if (!b || field < 0 || field >= NMF_FIELD_COUNT) { fail("nmf_field_ptr: bad field"); return nullptr; }
  if (width) *width = b->widths[field];
  return b->fields[field];
}

extern "C" int nmf_gather(nmf_batch* b, int field, const int32_t* ids_dev, int n_ids, int group, float* dst_dev, void* stream) {
  if (!b || field < 0 || field >= NMF_FIELD_COUNT) return fail("nmf_gather: bad field");
  if (n_ids <= 0) return 0;
  if (group < 1 || !ids_dev || !dst_dev) return fail("nmf_gather: bad arguments");
  DEVICE_GUARD(b);
  size_t total = (size_t)b->n_worlds * n_ids * group;
  int blocks = (int)((total + 255) / 256); if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(nmf::nmf_gather_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, b->fields[field],
                     b->widths[field], ids_dev, n_ids, group, dst_dev, b->n_worlds);
  HIP_OK(hipGetLastError());
  return 0;
}

extern "C" int nmf_scatter(nmf_batch* b, int field, const int32_t* ids_dev, int n_ids, const float* src_dev, void* stream) {
  if (!b || field < 0 || field >= NMF_FIELD_COUNT) return fail("nmf_scatter: bad field");
  if (n_ids <= 0) return 0;
  if (!ids_dev || !src_dev) return fail("nmf_scatter: bad arguments");
  DEVICE_GUARD(b);
  size_t total = (size_t)b->n_worlds * n_ids;
  int blocks = (int)((total + 255) / 256); if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(nmf::nmf_scatter_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, b->fields[field],
                     b->widths[field], ids_dev, n_ids, src_dev, b->n_worlds);
  HIP_OK(hipGetLastError());
  return 0;
}

extern "C" int nmf_pack_observations(nmf_batch* b, int n_joint, int n_act, float* out_dev, int row_stride, void* stream) {
  if (!b || !out_dev) return fail("nmf_pack_observations: null argument");
  const nmf_model* m = b->model;
  const int width = 2 * n_joint + n_act + 96;
  if (n_joint < 0 || n_joint > m->nv - 6 || n_act < 0 || n_act > m->nu || row_stride < width)
    return fail("nmf_pack_observations: need 0 <= n_joint <= nv - 6, 0 <= n_act <= nu and row_stride >= 2 n_joint + n_act + 96");
  DEVICE_GUARD(b);
  const size_t total = (size_t)b->n_worlds * width;
  int blocks = (int)((total + 255) / 256); if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(nmf::nmf_pack_obs_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, b->st.qpos, b->st.qvel,
                     b->st.actuator_force, b->st.sensordata, m->nq, m->nv, m->nu, n_joint, n_act, b->n_worlds, out_dev, row_stride);
  HIP_OK(hipGetLastError());
  return 0;
}

extern "C" int64_t nmf_step_count(const nmf_batch* b) { return b ? b->steps : 0; }

extern "C" int nmf_shader_clock(nmf_batch* b, double* hz_out, int reset) {
  if (!b || !hz_out) return fail("nmf_shader_clock: null argument");
  DEVICE_GUARD(b);
  unsigned long long v[2] = {0, 0};
  HIP_OK(hipDeviceSynchronize());
  HIP_OK(hipMemcpy(v, b->clock_probe_buf, sizeof(v), hipMemcpyDeviceToHost));
  *hz_out = v[1] ? 1e8 * (double)v[0] / (double)v[1] : 0.0;
  if (reset) HIP_OK(hipMemset(b->clock_probe_buf, 0, sizeof(v)));
  return 0;
}

extern "C" double nmf_time_launches(nmf_batch* b, const float* table_dev, int table_steps, int n_act,
                                    const int32_t* act_ids_dev, int n_steps, int reps, void* stream) {
  if (!b || reps <= 0 || n_steps <= 0) { fail("nmf_time_launches: bad arguments"); return -1.0; }
  hipStream_t s = (hipStream_t)stream;
  hipEvent_t e0, e1;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) { fail("hipEventCreate failed"); return -1.0; }
  (void)hipEventRecord(e0, s);
  int start = 0;
  for (int r = 0; r < reps; ++r) {
    int rc = table_dev ? nmf_step_replay(b, table_dev, table_steps, n_act, act_ids_dev, start, n_steps, stream)
                       : nmf_step(b, n_steps, stream);
    if (rc != 0) { (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); return -1.0; }
    start += n_steps;
  }
  (void)hipEventRecord(e1, s);
  if (hipEventSynchronize(e1) != hipSuccess) { fail("hipEventSynchronize failed"); return -1.0; }
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  return (double)ms / reps;
}

extern "C" size_t nmf_retina_plan_bytes(int n_pixels) {
  if (n_pixels <= 0 || n_pixels % 16) return 0;
  const size_t n_chunk = (size_t)n_pixels / 16;
  return n_chunk * 16 + ((n_chunk * 4 + 4 + 15) / 16) * 16;
}

extern "C" int nmf_retina_plan(const int16_t* id_map_dev, int n_pixels, void* plan_dev, void* stream) {
  if (!id_map_dev || !plan_dev) return fail("nmf_retina_plan: null buffer");
  if (n_pixels <= 0 || n_pixels % 16) return fail("nmf_retina_plan: n_pixels must be a positive multiple of 16");
  if (reinterpret_cast<uintptr_t>(plan_dev) & 15u) return fail("nmf_retina_plan: plan must be 16-byte aligned");
  const int n_chunk = n_pixels / 16;
  hipLaunchKernelGGL(nmf::nmf_retina_plan_kernel, dim3((unsigned)((n_chunk + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     id_map_dev, n_chunk, reinterpret_cast<nmf::u32x4*>(plan_dev));
  hipLaunchKernelGGL(nmf::nmf_retina_active_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, id_map_dev, n_chunk,
                     reinterpret_cast<int*>(static_cast<char*>(plan_dev) + (size_t)n_chunk * 16));
  HIP_OK(hipGetLastError());
  return 0;
}

extern "C" int nmf_retina_resample(const uint8_t* images_dev, const int16_t* id_map_dev, const void* plan_dev,
                                   const uint8_t* pale_dev, const float* inv_norm_dev, int n_images, int n_pixels,
                                   int n_ommatidia, float* out_dev, void* stream) {
  if (!images_dev || !id_map_dev || !pale_dev || !inv_norm_dev || !out_dev) return fail("nmf_retina_resample: null buffer");
  if (n_images <= 0) return 0;
  if (n_pixels <= 0 || n_ommatidia <= 0 || n_ommatidia > nmf::kMaxOmmatidia)
    return fail("nmf_retina_resample: need 0 < n_ommatidia <= 1024 and n_pixels > 0");
  if ((reinterpret_cast<uintptr_t>(images_dev) | reinterpret_cast<uintptr_t>(id_map_dev) | reinterpret_cast<uintptr_t>(plan_dev)) & 15u ||
      (n_pixels * 3) % 16)
    return fail("nmf_retina_resample: images / id map / plan must be 16-byte aligned and image size a multiple of 16 bytes");
  if (plan_dev && n_pixels % 1024 == 0)
    hipLaunchKernelGGL(nmf::nmf_retina_stream_kernel, dim3((unsigned)n_images), dim3(nmf::kRetinaThreads), 0, (hipStream_t)stream,
                       images_dev, id_map_dev, reinterpret_cast<const nmf::u32x4*>(plan_dev), pale_dev, inv_norm_dev, n_pixels,
                       n_ommatidia, out_dev);
  else
    hipLaunchKernelGGL(nmf::nmf_retina_kernel, dim3((unsigned)n_images), dim3(nmf::kRetinaThreads), 0, (hipStream_t)stream,
                       images_dev, id_map_dev, pale_dev, inv_norm_dev, n_pixels, n_ommatidia, out_dev);
  HIP_OK(hipGetLastError());
  return 0;
}

extern "C" size_t nmf_eye_params_size(void) { return sizeof(nmf_eye_params); }

// Visit plan of the eye renderer: the chunks (16 consecutive raw pixels) to render, sorted by 32 x 32-pixel tile and cut
// into groups of 64 (one wave's turn), each group with the bounding cone of its rays in the camera frame (axis, cos and
// sin of the half-angle) — so that a wave can decide per group what it can see at all.  Built on the host once per id map.
extern "C" void nmf_eye_plan_destroy(nmf_eye_plan* P) {
  if (!P) return;
  DeviceGuard guard(P->device);
  for (void* q : P->allocs) (void)hipFree(q);      // (hipFree waits for the kernels that read them)
  delete P;
}

// NOT stream-ordered: device-to-host copies of the id map / run plan / pale flags, host-side sorting, allocations and uploads — once
// per (id map, lens, ommatidia types).  Every failure path gives back what was allocated (nmf_eye_plan_destroy).
static int fill_eye_plan(nmf_eye_plan* Pp, const int16_t* id_map_dev, const void* plan_dev, const uint8_t* pale_dev, const float* inv_norm_dev, int h, int w, float fov_deg, int n_omm) {
  nmf_eye_plan& P = *Pp;
  auto dev_alloc = [&](size_t bytes) -> void* { void* q = nullptr; if (hipMalloc(&q, bytes) != hipSuccess) return nullptr; P.allocs.push_back(q); return q; };
  const int n_pix = h * w, n_chunk = n_pix / 16;
  std::vector<int16_t> ids((size_t)n_pix);
  HIP_OK(hipMemcpy(ids.data(), id_map_dev, sizeof(int16_t) * (size_t)n_pix, hipMemcpyDeviceToHost));
  const double half_fov = 0.5 * fov_deg * 3.14159265358979323846 / 180.0, inv_half_h = 2.0 / h, cx = 0.5 * w, cy = 0.5 * h;
  std::vector<double> ray((size_t)n_pix * 3);
  for (int i = 0; i < n_pix; ++i) {
    const int row = i / w, col = i - row * w;
    const double u = (col + 0.5 - cx) * inv_half_h, v = (row + 0.5 - cy) * inv_half_h;
    const double rho = std::sqrt(std::max(u * u + v * v, 1e-24)), th = rho * half_fov;
    ray[3 * (size_t)i] = std::sin(th) * u / rho; ray[3 * (size_t)i + 1] = -std::sin(th) * v / rho; ray[3 * (size_t)i + 2] = -std::cos(th);
  }
  // bounding cone (axis, cos of the half-angle) of a set of pixels
  auto cone_of = [&](const std::vector<size_t>& px, double margin, float* out) {
    double ax = 0, ay = 0, az = 0;
    for (size_t i : px) { ax += ray[3 * i]; ay += ray[3 * i + 1]; az += ray[3 * i + 2]; }
    const double nrm = std::sqrt(ax * ax + ay * ay + az * az);
    if (px.empty() || nrm < 1e-9) { out[0] = 0.f; out[1] = 0.f; out[2] = -1.f; out[3] = px.empty() ? 1.f : -1.f; return; }   // nothing / everything
    ax /= nrm; ay /= nrm; az /= nrm;
    double cmin = 1.0;
    for (size_t i : px) cmin = std::min(cmin, ax * ray[3 * i] + ay * ray[3 * i + 1] + az * ray[3 * i + 2]);
    out[0] = (float)ax; out[1] = (float)ay; out[2] = (float)az; out[3] = (float)std::max(-1.0, cmin - margin);
  };
  bool lens_ok = true;
  for (int i = 0; i < n_pix; ++i) lens_ok = lens_ok && ray[3 * (size_t)i + 2] < -std::cos(2.1);
  for (int mode = 0; mode < 2; ++mode) {
    // chunks that stay inside one image row, by tile; the ones that wrap to the next row (their rays lie at both image
    // edges: no common cone) in groups of their own, with one cone per piece
    std::vector<int> plain, wrapped;
    for (int ch = 0; ch < n_chunk; ++ch) {
      bool on = mode == 1;
      for (int k = 0; k < 16 && !on; ++k) on = (ids[(size_t)ch * 16 + k] & 0x7fff) != 0;
      if (!on) continue;
      ((ch * 16) / w == (ch * 16 + 15) / w ? plain : wrapped).push_back(ch);
    }
    auto key = [&](int ch) { const int row = (ch * 16) / w, col = ch * 16 - row * w; return ((long long)(row / 32) << 40) | ((long long)(col / 32) << 28) | ((long long)row << 12) | col; };
    std::sort(plain.begin(), plain.end(), [&](int a, int c) { return key(a) < key(c); });
    const int g_plain = (int)((plain.size() + 63) / 64), g_wrapped = (int)((wrapped.size() + 63) / 64), n_groups = g_plain + g_wrapped;
    std::vector<int> visit((size_t)n_groups * 64, -1);
    std::copy(plain.begin(), plain.end(), visit.begin());
    std::copy(wrapped.begin(), wrapped.end(), visit.begin() + (size_t)g_plain * 64);
    // per group: [0..3] cone of piece 0, [4..7] cone of piece 1 (wrapped groups), [8] 1 = two pieces; per slot: two cones
    std::vector<float> cones((size_t)n_groups * 12, 0.f), ccones(visit.size() * 8, 0.f);
    for (int g = 0; g < n_groups; ++g) {
      std::vector<size_t> gp[2];
      for (int l = 0; l < 64; ++l) {
        const size_t sl = (size_t)g * 64 + l;
        const int ch = visit[sl];
        std::vector<size_t> cp[2];
        if (ch >= 0) {
          const int row0 = (ch * 16) / w;
          for (int k = 0; k < 16; ++k) { const size_t i = (size_t)ch * 16 + k; const int pc = (int)(i / w) == row0 ? 0 : 1; cp[pc].push_back(i); gp[pc].push_back(i); }
        }
        cone_of(cp[0], 1e-5, &ccones[8 * sl]); cone_of(cp[1], 1e-5, &ccones[8 * sl + 4]);
      }
      cone_of(gp[0], 1e-4, &cones[(size_t)g * 12]); cone_of(gp[1], 1e-4, &cones[(size_t)g * 12 + 4]);
      cones[(size_t)g * 12 + 8] = g >= g_plain ? 1.f : 0.f;
      cones[(size_t)g * 12 + 9] = lens_ok ? 1.f : 0.f;      // [9]: every ray of the IMAGE within the range of the kernel's lens polynomials
    }
    void* pv = dev_alloc(sizeof(int) * visit.size()); void* pc = dev_alloc(sizeof(float) * cones.size()); void* pcc = dev_alloc(sizeof(float) * ccones.size());
    if (!pv || !pc || !pcc) return fail("nmf_eye_plan_create: out of device memory for the visit plan");
    HIP_OK(hipMemcpy(pv, visit.data(), sizeof(int) * visit.size(), hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(pc, cones.data(), sizeof(float) * cones.size(), hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(pcc, ccones.data(), sizeof(float) * ccones.size(), hipMemcpyHostToDevice));
    P.visit[mode] = (int*)pv; P.cones[mode] = (float*)pc; P.chunk_cones[mode] = (float*)pcc; P.n_groups[mode] = n_groups;
  }
  {
    // Sampled mode (nmf_eye_params::rays_per_ommatidium = 16): per ommatidium i the pixels of its cell in raster order,
    // P_0 .. P_{n-1}, and of those the 16 at indices floor((2 j + 1) n / 32), j = 0..15 (oracle/sensors_oracle.py::
    // sampled_pixels).  One lane per ray, an ommatidium's 16 rays in one DPP row.  The ommatidia are visited in 64 x 64
    // pixel tiles of their cells' centres, 16 (a 4 x 4 patch of the lattice) to a group: one bounding cone and one culling
    // for the group's 256 rays.  slot_omm[s] = the ommatidium in slot s.
    // Both lists carry what a ray would otherwise look up behind them (the sampled kernel waits for memory, not for arithmetic:
    // three dependent round trips per turn were visit -> the chunk's plan word, slot -> ommatidium -> pale): a visit entry is the
    // pixel | bit 30 "its chunk lies inside one image row and has a run plan" (the lens-polynomial path of the pixel-exact
    // kernel, nmf_eyes.hip); a slot entry is the ommatidium | bit 30 "pale".
    constexpr int K = nmf::kEyeRays, S = nmf::kEyeSlots;
    std::vector<uint32_t> planw((size_t)n_chunk * 4);
    std::vector<uint8_t> pale_h((size_t)n_omm);
    HIP_OK(hipDeviceSynchronize());        // (the caller's run plan was built on a stream of its own)
    HIP_OK(hipMemcpy(planw.data(), plan_dev, sizeof(uint32_t) * planw.size(), hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(pale_h.data(), pale_dev, pale_h.size(), hipMemcpyDeviceToHost));
    std::vector<std::vector<int>> cell((size_t)n_omm);
    for (int i = 0; i < n_pix; ++i) { const int id = ids[(size_t)i] & 0x7fff; if (id > 0 && id <= n_omm) cell[(size_t)id - 1].push_back(i); }
    std::vector<int> order((size_t)n_omm);
    std::vector<long long> keyv((size_t)n_omm);
    for (int o = 0; o < n_omm; ++o) {
      order[(size_t)o] = o;
      double r = 0, c = 0;
      for (int i : cell[(size_t)o]) { r += i / w; c += i % w; }
      const double n = std::max<size_t>(cell[(size_t)o].size(), 1);
      const int row = (int)(r / n), col = (int)(c / n);
      keyv[(size_t)o] = ((long long)(row / 64) << 40) | ((long long)(col / 64) << 28) | ((long long)row << 12) | col;
    }
    std::stable_sort(order.begin(), order.end(), [&](int a, int c) { return keyv[(size_t)a] < keyv[(size_t)c]; });
    const int n_groups = (n_omm + S - 1) / S;
    std::vector<int> visit((size_t)n_groups * S * K, -1), slots((size_t)n_groups * S, 0);
    for (int sl = 0; sl < n_omm; ++sl) {
      const int o = order[(size_t)sl];
      slots[(size_t)sl] = o | (pale_h[(size_t)o] ? (1 << 30) : 0);
      const long long n = (long long)cell[(size_t)o].size();
      for (int j = 0; j < K && n > 0; ++j) {
        const int px = cell[(size_t)o][(size_t)(((2 * j + 1) * n) / (2 * K))];
        const int pchunk = px >> 4;
        const bool one_row = (pchunk * 16) / w == (pchunk * 16 + 15) / w, planned = !(planw[(size_t)pchunk * 4 + 1] & 0x10000u);
        visit[(size_t)sl * K + j] = px | (one_row && planned ? (1 << 30) : 0);
      }
    }
    // ... and one cone per TURN (64 rays = four ommatidia): every ray tests all the capsules its culling lets through, so the
    // sampled kernel culls a second time per turn (`chunk_cones` of this mode: axis + cos of the half-angle, camera frame)
    std::vector<float> cones((size_t)n_groups * 12, 0.f), tcones((size_t)n_groups * (S / 4) * 4, 0.f);
    for (int g = 0; g < n_groups; ++g) {
      std::vector<size_t> gp;
      for (int l = 0; l < S * K; ++l) if (visit[(size_t)g * S * K + l] >= 0) gp.push_back((size_t)(visit[(size_t)g * S * K + l] & 0xffffff));
      cone_of(gp, 1e-4, &cones[(size_t)g * 12]);
      cones[(size_t)g * 12 + 9] = lens_ok ? 1.f : 0.f;
      for (int t = 0; t < S / 4; ++t) {
        std::vector<size_t> tp;
        for (int l = 64 * t; l < 64 * (t + 1); ++l) if (visit[(size_t)g * S * K + l] >= 0) tp.push_back((size_t)(visit[(size_t)g * S * K + l] & 0xffffff));
        cone_of(tp, 1e-4, &tcones[((size_t)g * (S / 4) + t) * 4]);
      }
    }
    void* pv = dev_alloc(sizeof(int) * visit.size()); void* pc = dev_alloc(sizeof(float) * cones.size());
    void* ps = dev_alloc(sizeof(int) * slots.size()); void* pt = dev_alloc(sizeof(float) * tcones.size());
    if (!pv || !pc || !ps || !pt) return fail("nmf_eye_plan_create: out of device memory for the sampling plan");
    HIP_OK(hipMemcpy(pt, tcones.data(), sizeof(float) * tcones.size(), hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(pv, visit.data(), sizeof(int) * visit.size(), hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(pc, cones.data(), sizeof(float) * cones.size(), hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(ps, slots.data(), sizeof(int) * slots.size(), hipMemcpyHostToDevice));
    P.visit[2] = (int*)pv; P.cones[2] = (float*)pc; P.chunk_cones[2] = (float*)pt; P.n_groups[2] = n_groups; P.slot_omm = (int*)ps;
  }
  // the plan's own copies of what the kernel reads from the caller's buffers
  {
    const size_t rbytes = nmf_retina_plan_bytes(n_pix);
    P.id_map = (int16_t*)dev_alloc(sizeof(int16_t) * (size_t)n_pix); P.rplan = dev_alloc(rbytes);
    P.pale = (uint8_t*)dev_alloc(((size_t)n_omm + 15) & ~(size_t)15); P.inv_norm = (float*)dev_alloc(sizeof(float) * (size_t)n_omm);
    if (!P.id_map || !P.rplan || !P.pale || !P.inv_norm) return fail("nmf_eye_plan_create: out of device memory for the plan's copies");
    HIP_OK(hipMemcpy(P.id_map, id_map_dev, sizeof(int16_t) * (size_t)n_pix, hipMemcpyDeviceToDevice));
    HIP_OK(hipMemcpy(P.rplan, plan_dev, rbytes, hipMemcpyDeviceToDevice));
    HIP_OK(hipMemcpy(P.pale, pale_dev, (size_t)n_omm, hipMemcpyDeviceToDevice));
    HIP_OK(hipMemcpy(P.inv_norm, inv_norm_dev, sizeof(float) * (size_t)n_omm, hipMemcpyDeviceToDevice));
    HIP_OK(hipDeviceSynchronize());
  }
  P.h = h; P.w = w; P.fov = fov_deg; P.n_omm = n_omm;
  return 0;
}

extern "C" nmf_eye_plan* nmf_eye_plan_create(const int16_t* id_map_dev, const void* plan_dev, const uint8_t* pale_dev, const float* inv_norm_dev,
                                             int height, int width, float fov_deg, int n_ommatidia, int device) {
  if (!id_map_dev || !plan_dev || !pale_dev || !inv_norm_dev) { fail("nmf_eye_plan_create: id map, plan, pale and inv_norm are required"); return nullptr; }
  if (height <= 0 || width <= 0 || (height * width) % 16) { fail("nmf_eye_plan_create: height * width must be a positive multiple of 16"); return nullptr; }
  if (n_ommatidia <= 0 || n_ommatidia > nmf::kMaxOmmatidia) { fail("nmf_eye_plan_create: need 0 < n_ommatidia <= 1024"); return nullptr; }
  if (!(fov_deg > 0.f) || fov_deg > 360.f) { fail("nmf_eye_plan_create: bad field of view"); return nullptr; }
  if (reinterpret_cast<uintptr_t>(plan_dev) & 15u) { fail("nmf_eye_plan_create: the run plan must be 16-byte aligned"); return nullptr; }
  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) != hipSuccess || device < 0 || device >= n_dev) { fail("nmf_eye_plan_create: no such device"); return nullptr; }
  DeviceGuard guard(device);
  nmf_eye_plan* P = new nmf_eye_plan();
  P->device = device;
  if (fill_eye_plan(P, id_map_dev, plan_dev, pale_dev, inv_norm_dev, height, width, fov_deg, n_ommatidia) != 0) {
    const std::string keep = g_err;
    nmf_eye_plan_destroy(P);
    g_err = keep;
    return nullptr;
  }
  return P;
}

// Pure stream-ordered work: argument checks on the host, one kernel launch.  Everything the kernel reads besides the batch's poses and
// the caller's sphere / capsule lists belongs to the plan.
extern "C" int nmf_eye_render_planned(nmf_batch* b, const nmf_eye_params* p, const nmf_eye_plan* plan, const float* spheres_dev,
                                      const int32_t* capsule_seg_dev, const float* capsule_geom_dev,
                                      uint8_t* frames_out_dev, float* omm_out_dev, void* stream) {
  if (!b || !p || !plan) return fail("nmf_eye_render: null batch / params / plan");
  if (plan->device != b->device) return fail("nmf_eye_render: the plan lives on another device than the batch");
  if (plan->h != p->height || plan->w != p->width || plan->fov != p->fov_deg) return fail("nmf_eye_render: the plan was made for another frame size / field of view");
  const int n_ommatidia = plan->n_omm;
  const int16_t* const id_map_dev = plan->id_map; const void* const plan_dev = plan->rplan; const uint8_t* const pale_dev = plan->pale; const float* const inv_norm_dev = plan->inv_norm;
  if (!frames_out_dev && !omm_out_dev) return fail("nmf_eye_render: nothing to write");
  if (p->height <= 0 || p->width <= 0 || (p->height * p->width) % 16) return fail("nmf_eye_render: height * width must be a positive multiple of 16");
  if (n_ommatidia <= 0 || n_ommatidia > nmf::kMaxOmmatidia) return fail("nmf_eye_render: need 0 < n_ommatidia <= 1024");
  if (p->n_spheres < 0 || p->n_spheres > nmf::kMaxSpheres || (p->n_spheres > 0 && !spheres_dev)) return fail("nmf_eye_render: bad sphere list");
  if (p->n_capsules < 0 || p->n_capsules > nmf::kMaxCaps || (p->n_capsules > 0 && (!capsule_seg_dev || !capsule_geom_dev)))
    return fail("nmf_eye_render: bad body-capsule list (at most 64)");
  if (!(p->checker_size > 0.f) || !(p->fov_deg > 0.f) || p->fov_deg > 360.f) return fail("nmf_eye_render: bad checker size / field of view");
  const nmf_model* m = b->model;
  DEVICE_GUARD(b);
  if (b->dm.plane[0] != 0.f || b->dm.plane[1] != 0.f || b->dm.plane[2] != 1.f) return fail("nmf_eye_render: the ground plane must be z-up");
  if ((reinterpret_cast<uintptr_t>(plan_dev) | reinterpret_cast<uintptr_t>(frames_out_dev)) & 15u)
    return fail("nmf_eye_render: plan / frames must be 16-byte aligned");
  nmf::EyeArgs A{};
  A.height = p->height; A.width = p->width;
  A.half_fov = 0.5f * p->fov_deg * 3.14159265358979323846f / 180.f;
  for (int e = 0; e < 2; ++e) {
    if (p->eye_seg[e] < 0 || p->eye_seg[e] >= m->nseg) return fail("nmf_eye_render: eye segment out of range");
    A.eye_seg[e] = p->eye_seg[e];
    for (int i = 0; i < 3; ++i) A.rel_pos[e][i] = p->rel_pos[e][i];
    double w = p->rel_quat[e][0], x = p->rel_quat[e][1], y = p->rel_quat[e][2], z = p->rel_quat[e][3];
    const double n = std::sqrt(w * w + x * x + y * y + z * z);
    if (!(n > 0.0)) return fail("nmf_eye_render: zero camera quaternion");
    w /= n; x /= n; y /= n; z /= n;
    const double R[9] = {1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
                         2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                         2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)};
    for (int i = 0; i < 9; ++i) A.rel_mat[e][i] = (float)R[i];
  }
  A.checker_size = p->checker_size; A.ground_z = b->dm.plane[3];
  A.n_spheres = p->n_spheres; A.sphere_stride = p->spheres_per_world ? 4 * p->n_spheres : 0;
  A.n_caps = p->n_capsules;
  A.terrain_kind = p->terrain_relief ? b->dm.terrain_type : 0;          // the relief the physics of this batch collides with
  for (int k = 0; k < 5; ++k) A.terrain[k] = b->dm.terrain[k];
  for (int c = 0; c < 4; ++c) {
    A.rgb[0][c] = c < 3 ? p->sky_rgb[c] : 0; A.rgb[1][c] = c < 3 ? p->ground_rgb[0][c] : 0; A.rgb[2][c] = c < 3 ? p->ground_rgb[1][c] : 0;
    A.rgb[3][c] = c < 3 ? p->wall_rgb[c] : 0; A.rgb[4][c] = c < 3 ? p->body_rgb[c] : 0;
    for (int s = 0; s < nmf::kMaxSpheres; ++s) A.rgb[5 + s][c] = c < 3 ? p->sphere_rgb[s][c] : 0;
  }
  if (p->rays_per_ommatidium != 0 && p->rays_per_ommatidium != nmf::kEyeRays) return fail("nmf_eye_render: rays_per_ommatidium must be 0 (every pixel) or 16");
  if (p->rays_per_ommatidium != 0 && frames_out_dev) return fail("nmf_eye_render: the sampled mode renders no frames (rays_per_ommatidium = 0 does)");
  A.sampled = p->rays_per_ommatidium;
  const int mode = A.sampled ? 2 : frames_out_dev ? 1 : 0;
#define NMF_EYE_LAUNCH(SAMPLED, RELIEF, FRAMES)                                                                                      \
  hipLaunchKernelGGL((nmf::nmf_eye_kernel<SAMPLED, RELIEF, FRAMES>), dim3((unsigned)(2 * b->n_worlds)), dim3(nmf::kEyeThreads), 0, (hipStream_t)stream, A, \
                     b->st.seg_xpos, b->st.seg_xquat, m->nseg, spheres_dev ? spheres_dev : b->st.seg_xpos,                           \
                     capsule_seg_dev, capsule_geom_dev, reinterpret_cast<const nmf::u32x4*>(plan_dev),                               \
                     plan->visit[mode], plan->cones[mode], reinterpret_cast<const float4*>(plan->chunk_cones[mode]), plan->n_groups[mode], \
                     id_map_dev, plan->slot_omm, pale_dev, inv_norm_dev, n_ommatidia, frames_out_dev, omm_out_dev)
  if (A.terrain_kind != 0) { if (A.sampled) NMF_EYE_LAUNCH(true, true, false); else if (frames_out_dev) NMF_EYE_LAUNCH(false, true, true); else NMF_EYE_LAUNCH(false, true, false); }
  else { if (A.sampled) NMF_EYE_LAUNCH(true, false, false); else if (frames_out_dev) NMF_EYE_LAUNCH(false, false, true); else NMF_EYE_LAUNCH(false, false, false); }
#undef NMF_EYE_LAUNCH
  HIP_OK(hipGetLastError());
  return 0;
}

// The entry without a plan handle: plans are built on first use and kept per batch (up to four, least recently used first out), keyed
// on the ADDRESSES of the four buffers + shape + lens.  The first call with a new key is therefore NOT stream-ordered (see
// nmf_eye_plan_create) and must not sit inside a stream capture, and a caller that rewrites a buffer in place must make a new
// plan (nmf_eye_plan_create) — include/nmf.h says so at the declaration.
extern "C" int nmf_eye_render(nmf_batch* b, const nmf_eye_params* p, const float* spheres_dev, const int32_t* capsule_seg_dev,
                              const float* capsule_geom_dev, const int16_t* id_map_dev,
                              const void* plan_dev, const uint8_t* pale_dev, const float* inv_norm_dev, int n_ommatidia,
                              uint8_t* frames_out_dev, float* omm_out_dev, void* stream) {
  if (!b || !p) return fail("nmf_eye_render: null batch / params");
  if (!id_map_dev || !plan_dev || !pale_dev || !inv_norm_dev) return fail("nmf_eye_render: id map, plan, pale and inv_norm are required");
  auto& L = b->eye_plans;
  nmf_eye_plan* plan = nullptr;
  for (size_t i = 0; i < L.size(); ++i) {
    nmf_eye_plan* q = L[i];
    if (q->key[0] == id_map_dev && q->key[1] == plan_dev && q->key[2] == pale_dev && q->key[3] == inv_norm_dev && q->h == p->height && q->w == p->width &&
        q->fov == p->fov_deg && q->n_omm == n_ommatidia) { plan = q; L.erase(L.begin() + (long)i); L.push_back(q); break; }
  }
  if (!plan) {
    plan = nmf_eye_plan_create(id_map_dev, plan_dev, pale_dev, inv_norm_dev, p->height, p->width, p->fov_deg, n_ommatidia, b->device);
    if (!plan) return -1;
    plan->key[0] = id_map_dev; plan->key[1] = plan_dev; plan->key[2] = pale_dev; plan->key[3] = inv_norm_dev;
    if (L.size() >= 4) { nmf_eye_plan_destroy(L.front()); L.erase(L.begin()); }
    L.push_back(plan);
  }
  return nmf_eye_render_planned(b, p, plan, spheres_dev, capsule_seg_dev, capsule_geom_dev, frames_out_dev, omm_out_dev, stream);
}

extern "C" int nmf_odor_intensity(nmf_batch* b, const int32_t* sensor_seg_dev, const float* sensor_rel_dev, int n_sensors,
                                  const float* source_pos_dev, const float* source_peak_dev, int n_sources, int n_dims,
                                  float* out_dev, void* stream) {
  if (!b) return fail("nmf_odor_intensity: null batch");
  if (n_sensors <= 0 || n_sources < 0 || n_dims <= 0 || !sensor_seg_dev || !sensor_rel_dev || !out_dev)
    return fail("nmf_odor_intensity: bad arguments");
  int total = b->n_worlds * n_sensors;
  DEVICE_GUARD(b);
  hipLaunchKernelGGL(nmf::nmf_odor_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     b->st.seg_xpos, b->st.seg_xquat, b->model->nseg, sensor_seg_dev, sensor_rel_dev, n_sensors,
                     source_pos_dev, source_peak_dev, n_sources, n_dims, out_dev, b->n_worlds);
  HIP_OK(hipGetLastError());
  return 0;
}

extern "C" int nmf_replay_resample(const float* clip_dev, int n_frames, int n_cols, double fps, double out_dt,
                                   const double* sg_taps_dev, int window, int n_out, float* out_dev, void* stream) {
  if (!clip_dev || !sg_taps_dev || !out_dev) return fail("nmf_replay_resample: null buffer");
  if (n_frames < 6 || n_frames > nmf::kReplayMaxFrames) return fail("nmf_replay_resample: need 6 <= n_frames <= 1536");
  if (window < 3 || !(window & 1) || window > n_frames) return fail("nmf_replay_resample: the filter window must be odd, >= 3 and <= n_frames");
  if (n_cols <= 0 || n_out <= 0 || !(fps > 0.0) || !(out_dt > 0.0)) return fail("nmf_replay_resample: bad sizes / rates");
  hipLaunchKernelGGL(nmf::nmf_replay_resample_kernel, dim3((unsigned)n_cols), dim3(nmf::kReplayThreads), 0, (hipStream_t)stream,
                     clip_dev, n_frames, n_cols, fps, out_dt, sg_taps_dev, window, n_out, out_dev);
  HIP_OK(hipGetLastError());
  return 0;
}

#ifdef NMF_SCHED_TRACE
// diagnostic build only: per-workgroup schedule trace of the last stepping launch (see nmf_step.hip)
extern "C" int nmf_debug_sched_trace(unsigned long long* out, int n_groups) {
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(nmf::g_sched_trace), sizeof(unsigned long long) * 8 * (size_t)std::min(n_groups, 4096)) != hipSuccess) return -1;
  return 0;
}
#endif

#ifdef NMF_STAGE_PROFILE
// diagnostic build only: cumulative s_memtime cycles per pipeline stage of wave 0
extern "C" int nmf_debug_stage_cycles(unsigned long long* out, int n, int reset) {
  unsigned long long host[NMF_NSTAGE] = {};
  if (hipMemcpyFromSymbol(host, HIP_SYMBOL(nmf::g_stage_cycles), sizeof(host)) != hipSuccess) return -1;
  for (int i = 0; i < n && i < NMF_NSTAGE; ++i) out[i] = host[i];
  if (reset) { unsigned long long z[NMF_NSTAGE] = {}; (void)hipMemcpyToSymbol(HIP_SYMBOL(nmf::g_stage_cycles), z, sizeof(z)); }
  return 0;
}
#endif
#ifdef NMF_EYE_STATS
extern "C" int nmf_debug_eye_stats(unsigned long long* out) {
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(nmf::g_eye_stats), sizeof(unsigned long long) * 8) != hipSuccess) return -1;
  unsigned long long z[8] = {};
  (void)hipMemcpyToSymbol(HIP_SYMBOL(nmf::g_eye_stats), z, sizeof(z));
  return 0;
}
#endif
