// nmf_sensors.hip — vision and olfaction sensor kernels (gfx950).
//
// The reference snapshot (flygym 2.0.1) ships only the *constants* of these sensors
// (src/flygym/assets/model/legacy/flygym1_config.yaml:141-192: 721 ommatidia per eye, 512x450 raw image,
// eye-camera poses; odor sensor sites on the rostrum and the funiculi) — no code and no id-map files
// (SURVEY §0.3, §8 a18/a19).  The semantics below are therefore build-defined and pinned
// by the numpy oracle in oracle/sensors_oracle.py.
//
// Retina resample: every pixel of a raw eye image belongs to at most one ommatidium (typed id map: bits 0..14
// = id, 0 = none; bit 15 = the ommatidium is pale, so no per-pixel type lookup is needed);
// an ommatidium is "yellow" (reads the green channel) or "pale" (reads the blue channel); its reading
// is the mean of that channel over its pixels, scaled to [0, 1], stored in channel 0 (yellow) or 1
// (pale) of out[image][ommatidium][2], the other channel being 0.
//
// This kernel IS HBM-bound: 3 bytes in per pixel (691 200 B per eye frame) against 5.8 KB out.  One
// workgroup per image; each thread streams 16-pixel chunks as three 16-byte loads (+ two for the id
// map, which all images share and L2 keeps), folds runs of equal ids in registers and flushes them
// with integer LDS atomics — exact and order-independent, hence bit-reproducible.
#include "nmf_device.h"

namespace nmf {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int kRetinaThreads = 512;
constexpr int kMaxOmmatidia = 1024;

__global__ void __launch_bounds__(kRetinaThreads)
nmf_retina_kernel(const uint8_t* __restrict__ images, const int16_t* __restrict__ id_map,
                  const uint8_t* __restrict__ pale, const float* __restrict__ inv_norm, int n_pix, int n_omm,
                  float* __restrict__ out) {
  __shared__ unsigned int acc[kMaxOmmatidia];
  const int img = blockIdx.x;
  for (int i = threadIdx.x; i < n_omm; i += kRetinaThreads) acc[i] = 0u;
  __syncthreads();
  const uint8_t* src = images + (size_t)img * n_pix * 3;
  const int n_chunk = n_pix / 16;                       // 16 pixels = 48 image bytes + 32 id bytes
  // two chunks per iteration: ten 16-byte loads in flight per thread before any use
  auto fold = [&](const u32x4 a, const u32x4 b, const u32x4 c, const u32x4 i0, const u32x4 i1) {
    const unsigned int w[12] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, c.y, c.z, c.w};
    const unsigned int iw[8] = {i0.x, i0.y, i0.z, i0.w, i1.x, i1.y, i1.z, i1.w};
    int cur = 0; unsigned int sum = 0u;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const unsigned int tid = (iw[k >> 1] >> ((k & 1) * 16)) & 0xffffu;   // bit 15: pale (reads blue)
      const int id = (int)(tid & 0x7fffu);
      // bytes of pixel k: 3k (R), 3k+1 (G), 3k+2 (B)
      const unsigned int g = (w[(3 * k + 1) >> 2] >> (((3 * k + 1) & 3) * 8)) & 0xffu;
      const unsigned int bl = (w[(3 * k + 2) >> 2] >> (((3 * k + 2) & 3) * 8)) & 0xffu;
      if (id != cur) {
        if (cur > 0) atomicAdd(&acc[cur - 1], sum);
        cur = id; sum = 0u;
      }
      if (id > 0) sum += (tid & 0x8000u) ? bl : g;
    }
    if (cur > 0) atomicAdd(&acc[cur - 1], sum);
  };
  int ch = threadIdx.x;
  for (; ch + kRetinaThreads < n_chunk; ch += 2 * kRetinaThreads) {
    const u32x4* p0 = reinterpret_cast<const u32x4*>(src + (size_t)ch * 48);
    const u32x4* p1 = reinterpret_cast<const u32x4*>(src + (size_t)(ch + kRetinaThreads) * 48);
    // image bytes are read exactly once: non-temporal; the id map is shared by every image: cached
    const u32x4 a0 = __builtin_nontemporal_load(p0), b0 = __builtin_nontemporal_load(p0 + 1), c0 = __builtin_nontemporal_load(p0 + 2);
    const u32x4 a1 = __builtin_nontemporal_load(p1), b1 = __builtin_nontemporal_load(p1 + 1), c1 = __builtin_nontemporal_load(p1 + 2);
    const u32x4* q0 = reinterpret_cast<const u32x4*>(id_map + (size_t)ch * 16);
    const u32x4* q1 = reinterpret_cast<const u32x4*>(id_map + (size_t)(ch + kRetinaThreads) * 16);
    const u32x4 i00 = q0[0], i01 = q0[1], i10 = q1[0], i11 = q1[1];
    fold(a0, b0, c0, i00, i01);
    fold(a1, b1, c1, i10, i11);
  }
  for (; ch < n_chunk; ch += kRetinaThreads) {
    const u32x4* p = reinterpret_cast<const u32x4*>(src + (size_t)ch * 48);
    const u32x4* q = reinterpret_cast<const u32x4*>(id_map + (size_t)ch * 16);
    fold(__builtin_nontemporal_load(p), __builtin_nontemporal_load(p + 1), __builtin_nontemporal_load(p + 2), q[0], q[1]);
  }
  // tail pixels (n_pix not a multiple of 16)
  for (int px = n_chunk * 16 + threadIdx.x; px < n_pix; px += kRetinaThreads) {
    const unsigned int tid = (unsigned short)id_map[px];
    const int id = (int)(tid & 0x7fffu);
    if (id > 0) atomicAdd(&acc[id - 1], (unsigned int)src[(size_t)px * 3 + ((tid & 0x8000u) ? 2 : 1)]);
  }
  __syncthreads();
  float* dst = out + (size_t)img * n_omm * 2;
  for (int i = threadIdx.x; i < n_omm; i += kRetinaThreads) {
    const float v = (float)acc[i] * inv_norm[i];
    const bool p = pale[i] != 0;
    dst[2 * i] = p ? 0.f : v;
    dst[2 * i + 1] = p ? v : 0.f;
  }
}

// ---- streaming variant (the one used for real eye frames) -------------------------------------------------------
// The run structure of a 16-pixel chunk depends on the id map only, so it is planned once (nmf_retina_plan_kernel):
// a chunk is at most three runs a | b | c of equal typed id; the plan holds the ids, the pixel masks of "in run a"
// and "in run a or b", and the pale-pixel mask (16 bytes per chunk instead of 32 bytes of ids).
//   * loads: a wave fetches 64 chunks = 3072 contiguous bytes with three 16-byte non-temporal loads per lane (lane l
//     takes bytes 16 l of each KiB: fully coalesced, ≈7 TB/s pattern — per-lane 48-byte chunks reach only 4.6), parks
//     them in its 3 KiB LDS stage and reads back its own 48-byte chunk (ds_read_b128 at stride 48 B is bank-conflict
//     free: 12 l mod 64 visits every bank quad once per 16 lanes);
//   * arithmetic: 4 pixels = 12 bytes = 3 dwords; two v_perm_b32 gather their four G bytes, two more the B bytes,
//     v_bfi picks B for pale pixels, and v_dot4_u32_u8 against 0/1 byte masks gives the prefix sums
//     P(n1), P(n2), P(16) — 4.5 instructions per pixel instead of ≈20;
//   * three LDS atomics per chunk (run sums), integer: exact and order independent.
// Chunks with more than three runs (flag in the plan) take the per-pixel path.  Needs n_pix % 1024 == 0.
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__global__ void nmf_retina_plan_kernel(const int16_t* __restrict__ id_map, int n_chunk, u32x4* __restrict__ plan) {
  const int ch = blockIdx.x * blockDim.x + threadIdx.x;
  if (ch >= n_chunk) return;
  unsigned int tid[16];
  for (int k = 0; k < 16; ++k) tid[k] = (unsigned short)id_map[(size_t)ch * 16 + k];
  int n1 = 16, n2 = 16;
  for (int k = 15; k >= 1; --k) if (tid[k] != tid[0]) n1 = k;
  const unsigned int ida = tid[0], idb = n1 < 16 ? tid[n1] : 0u;
  for (int k = 15; k > n1; --k) if (tid[k] != idb) n2 = k;
  const unsigned int idc = n2 < 16 ? tid[n2] : 0u;
  bool bad = false; unsigned int palebits = 0u;
  for (int k = 0; k < 16; ++k) {
    if (k >= n2 && tid[k] != idc) bad = true;
    palebits |= ((tid[k] >> 15) & 1u) << k;
  }
  const unsigned int m1 = (1u << n1) - 1u, m2 = (1u << n2) - 1u;
  plan[ch] = u32x4{ida | (idb << 16), idc | ((bad ? 1u : 0u) << 16), palebits, m1 | (m2 << 16)};
}

// The chunks that touch at least one ommatidium, in ascending order, after the plan entries:
// plan blob = [n_chunk x 16 B run plan][n_chunk x int32 active chunk indices][int32 count, padded to 16 B].
// One workgroup; order-preserving compaction by ballot prefix sums.
__global__ void __launch_bounds__(256) nmf_retina_active_kernel(const int16_t* __restrict__ id_map, int n_chunk, int* __restrict__ list) {
  __shared__ int wave_cnt[4];
  __shared__ int base;
  if (threadIdx.x == 0) base = 0;
  __syncthreads();
  for (int c0 = 0; c0 < n_chunk; c0 += 256) {
    const int ch = c0 + threadIdx.x;
    bool on = false;
    if (ch < n_chunk)
      for (int k = 0; k < 16; ++k) on |= (id_map[(size_t)ch * 16 + k] & 0x7fff) != 0;
    const unsigned long long bal = __ballot(on);
    const int wv = threadIdx.x >> 6, ln = threadIdx.x & 63;
    if (ln == 0) wave_cnt[wv] = __popcll(bal);
    __syncthreads();
    int off = base;
    for (int q = 0; q < wv; ++q) off += wave_cnt[q];
    if (on) list[off + __popcll(bal & ((1ull << ln) - 1ull))] = ch;
    __syncthreads();
    if (threadIdx.x == 0) base += wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
    __syncthreads();
  }
  if (threadIdx.x == 0) list[n_chunk] = base;
}

// 4 bits -> 4 bytes of 0/1
__device__ __forceinline__ unsigned int nib_to_bytes(unsigned int nib) { return (nib * 0x00204081u) & 0x01010101u; }

__global__ void __launch_bounds__(kRetinaThreads)
nmf_retina_stream_kernel(const uint8_t* __restrict__ images, const int16_t* __restrict__ id_map, const u32x4* __restrict__ plan,
                         const uint8_t* __restrict__ pale, const float* __restrict__ inv_norm, int n_pix, int n_omm,
                         float* __restrict__ out) {
  constexpr int kWaves = kRetinaThreads / 64;
  __shared__ unsigned int acc[kMaxOmmatidia];
  __shared__ u32x4 stage[kWaves][2][192];
  const int img = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < n_omm; i += kRetinaThreads) acc[i] = 0u;
  __syncthreads();
  const u32x4* src = reinterpret_cast<const u32x4*>(images + (size_t)img * n_pix * 3);
  const int n_group = n_pix / 1024;                    // 64 chunks of 16 pixels
  auto consume = [&](const int ch, const u32x4 pl, const u32x4 x0, const u32x4 x1, const u32x4 x2) {
    const unsigned int w[12] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w, x2.x, x2.y, x2.z, x2.w};
    if (!(pl.y & 0x10000u)) {
      unsigned int tot = 0u, sA = 0u, sAB = 0u;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const unsigned int d0 = w[3 * g], d1 = w[3 * g + 1], d2 = w[3 * g + 2];
        // memory bytes of the 4 pixels: d0 = R0 G0 B0 R1, d1 = G1 B1 R2 G2, d2 = B2 R3 G3 B3
        const unsigned int G4 = __builtin_amdgcn_perm(d2, __builtin_amdgcn_perm(d1, d0, 0x0c070401u), 0x06020100u);
        const unsigned int B4 = __builtin_amdgcn_perm(d2, __builtin_amdgcn_perm(d1, d0, 0x0c0c0502u), 0x07040100u);
        const unsigned int P4 = nib_to_bytes((pl.z >> (4 * g)) & 0xfu) * 0xffu;
        const unsigned int V4 = (B4 & P4) | (G4 & ~P4);
        tot = __builtin_amdgcn_udot4(V4, 0x01010101u, tot, false);
        sA = __builtin_amdgcn_udot4(V4, nib_to_bytes((pl.w >> (4 * g)) & 0xfu), sA, false);
        sAB = __builtin_amdgcn_udot4(V4, nib_to_bytes((pl.w >> (16 + 4 * g)) & 0xfu), sAB, false);
      }
      const int ia = (int)(pl.x & 0x7fffu), ib = (int)((pl.x >> 16) & 0x7fffu), ic = (int)(pl.y & 0x7fffu);
      if (ia > 0) atomicAdd(&acc[ia - 1], sA);
      if (ib > 0) atomicAdd(&acc[ib - 1], sAB - sA);
      if (ic > 0) atomicAdd(&acc[ic - 1], tot - sAB);
    } else {
      const u32x4* qd = reinterpret_cast<const u32x4*>(id_map + (size_t)ch * 16);
      const u32x4 i0 = qd[0], i1 = qd[1];
      const unsigned int iw[8] = {i0.x, i0.y, i0.z, i0.w, i1.x, i1.y, i1.z, i1.w};
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        const unsigned int tid = (iw[k >> 1] >> ((k & 1) * 16)) & 0xffffu;
        const int id = (int)(tid & 0x7fffu);
        const unsigned int g = (w[(3 * k + 1) >> 2] >> (((3 * k + 1) & 3) * 8)) & 0xffu;
        const unsigned int bl = (w[(3 * k + 2) >> 2] >> (((3 * k + 2) & 3) * 8)) & 0xffu;
        if (id > 0) atomicAdd(&acc[id - 1], (tid & 0x8000u) ? bl : g);
      }
    }
  };
  // two groups per iteration: six 16-byte loads in flight per lane
  int gidx = wave;
  for (; gidx + kWaves < n_group; gidx += 2 * kWaves) {
    const u32x4* p = src + (size_t)gidx * 192 + lane;
    const u32x4* q = p + (size_t)kWaves * 192;
    const u32x4 a0 = __builtin_nontemporal_load(p), b0 = __builtin_nontemporal_load(p + 64), c0 = __builtin_nontemporal_load(p + 128);
    const u32x4 a1 = __builtin_nontemporal_load(q), b1 = __builtin_nontemporal_load(q + 64), c1 = __builtin_nontemporal_load(q + 128);
    const int ch0 = gidx * 64 + lane, ch1 = ch0 + kWaves * 64;
    const u32x4 pl0 = plan[ch0], pl1 = plan[ch1];
    stage[wave][0][lane] = a0; stage[wave][0][64 + lane] = b0; stage[wave][0][128 + lane] = c0;
    stage[wave][1][lane] = a1; stage[wave][1][64 + lane] = b1; stage[wave][1][128 + lane] = c1;
    wave_sync();
    const u32x4 x0 = stage[wave][0][3 * lane], x1 = stage[wave][0][3 * lane + 1], x2 = stage[wave][0][3 * lane + 2];
    const u32x4 y0 = stage[wave][1][3 * lane], y1 = stage[wave][1][3 * lane + 1], y2 = stage[wave][1][3 * lane + 2];
    wave_sync();
    consume(ch0, pl0, x0, x1, x2);
    consume(ch1, pl1, y0, y1, y2);
  }
  for (; gidx < n_group; gidx += kWaves) {
    const u32x4* p = src + (size_t)gidx * 192 + lane;
    const u32x4 a0 = __builtin_nontemporal_load(p), b0 = __builtin_nontemporal_load(p + 64), c0 = __builtin_nontemporal_load(p + 128);
    const int ch0 = gidx * 64 + lane;
    const u32x4 pl0 = plan[ch0];
    stage[wave][0][lane] = a0; stage[wave][0][64 + lane] = b0; stage[wave][0][128 + lane] = c0;
    wave_sync();
    const u32x4 x0 = stage[wave][0][3 * lane], x1 = stage[wave][0][3 * lane + 1], x2 = stage[wave][0][3 * lane + 2];
    wave_sync();
    consume(ch0, pl0, x0, x1, x2);
  }
  __syncthreads();
  float* dst = out + (size_t)img * n_omm * 2;
  for (int i = threadIdx.x; i < n_omm; i += kRetinaThreads) {
    const float v = (float)acc[i] * inv_norm[i];
    const bool pp = pale[i] != 0;
    dst[2 * i] = pp ? 0.f : v;
    dst[2 * i + 1] = pp ? v : 0.f;
  }
}

// Odor intensity at the fly's odor sensors:  out[w][d][k] = sum_s peak[s][d] / |x_sensor(w,k) - x_source(s)|^2
// (inverse-square diffusion, the flygym 1.x OdorArena default).  Sensor k sits at seg_xpos + R(seg_xquat) rel_pos
// of its parent segment.  One thread per (world, sensor).
__global__ void nmf_odor_kernel(const float* __restrict__ seg_xpos, const float* __restrict__ seg_xquat, int nseg,
                                const int* __restrict__ sensor_seg, const float* __restrict__ sensor_rel, int n_sensor,
                                const float* __restrict__ src_pos, const float* __restrict__ src_peak, int n_src, int n_dim,
                                float* __restrict__ out, int n_worlds) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_worlds * n_sensor) return;
  const int w = t / n_sensor, k = t % n_sensor, sg = sensor_seg[k];
  const float* xp = seg_xpos + ((size_t)w * nseg + sg) * 3;
  const Q4 q = ldq(seg_xquat + ((size_t)w * nseg + sg) * 4);
  float R[9];
  qmat(R, q);
  const V3 p = ld3(xp) + mat_vec(R, ld3(sensor_rel + 3 * k));
  for (int d = 0; d < n_dim; ++d) {
    float acc = 0.f;
    for (int s = 0; s < n_src; ++s) {
      const V3 e = p - ld3(src_pos + 3 * s);
      acc += src_peak[s * n_dim + d] / dot(e, e);
    }
    out[((size_t)w * n_dim + d) * n_sensor + k] = acc;
  }
}

}  // namespace nmf
