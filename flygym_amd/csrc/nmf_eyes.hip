// nmf_eyes.hip — compound-eye renderer fused with the ommatidia resample (gfx950).  SURVEY §8 (f) rank 2.
//
// The reference renders eye images with MuJoCo's OpenGL renderer / MJWarp's batch renderer and flygym 1.x then
// resampled them to ommatidia; this snapshot has neither the eye cameras nor the resample (only the constants in
// src/flygym/assets/model/legacy/flygym1_config.yaml:141-173), so the camera model and the scene are build-defined
//  and pinned by the numpy oracle oracle/sensors_oracle.py::render_eye_frames.
//
// One workgroup per (world, eye); a wave takes GROUPS of 64 chunks that form a compact patch of the image (visit plan built
// once per id map by nmf_capi.hip: chunks sorted by 32 x 32-pixel tile, each group with the bounding cone of its rays in
// the camera frame).  Per group the wave decides once — lane = object — which body capsules and spheres its cone can see
// and whether any ray of it points below the horizon; most groups see sky or ground only and skip the rest.  (Round 3
// tested all 55 capsules per lane and chunk: 41 of ~230 instructions per ray, and cast every ray against the ground.)
// A lane owns 16 consecutive raw pixels (the resample's chunk, same run plan):
// per pixel it builds the equidistant-fisheye ray, rotates it into the world, intersects the ground — the flat plane
// (checker texture) or, on the build-defined terrains, the relief the physics collides with (constant-height cells of
// h(x, y) followed cell by cell, side walls included) — the spheres and the fly's own body (the visible segments as
// capsules, reference scene: warp/rendering.py:385-441 renders the whole model; flygym 1.x hid the segments around the
// eyes, legacy/flygym1_config.yaml:147-161), takes the green or blue byte of the nearest hit's material and adds it to
// the chunk's run sums; three integer LDS atomics per chunk.  The 1.4 MB raw frame per fly never exists unless the caller asks for it
// (frames_out, for inspection and for the parity tests): compute-bound instead of HBM-bound.
#include "nmf_device.h"

// The two instantiations of the kernel below (pixel-exact, sampled) must colour a pixel identically — the sampled mode's readings
// equal its specification applied to the pixel-exact mode's frames bit for bit (tests/test_sensors.py) — so the ray and scene
// arithmetic may not be re-associated differently in the two contexts (the library is built with -fassociative-math).  Fused
// multiply-adds stay (contraction does not depend on the surrounding code).
#pragma clang fp reassociate(off)

namespace nmf {

#ifdef NMF_EYE_STATS
__device__ unsigned long long g_eye_stats[8];
#endif
constexpr int kEyeThreads = 256;
constexpr int kEyeRays = 16;        // rays per ommatidium of the sampled mode: one DPP row of lanes
constexpr int kEyeSlots = 16;       // ommatidia per group of the sampled mode (a 4 x 4 patch of the lattice: one culling for 256 rays)
constexpr int kMaxSpheres = 8;
constexpr int kMaxCaps = 64;            // capsules of the fly's own body an eye can see
constexpr int kMaxTerrainCells = 64;    // cells a ray is followed through the relief before the far field is taken as flat
constexpr float kTerrainEps = 1e-4f;    // the cell a ray is in at parameter t holds its point at t + eps (mm)
constexpr float kTerrainWallTol = 1e-4f;   // entered through the side wall: this far below the cell's level (levels differ by >= 0.3 mm)

struct EyeArgs {
  int height, width;
  float half_fov;          // polar angle (rad) of the ray through the middle of the top / bottom image edge
  int eye_seg[2];
  float rel_pos[2][3];
  float rel_mat[2][9];     // camera axes in the parent segment frame (columns: right, up, back)
  float checker_size, ground_z;
  int n_spheres, sphere_stride;      // floats between consecutive worlds in `spheres` (0: shared by all worlds)
  int terrain_kind;                  // 0 flat plane, 1 gapped, 2 blocks, 3 mixed (flygym_amd/compose/world.py)
  float terrain[5];                  // its parameters + highest level
  int n_caps;                        // body capsules (cap_seg / cap_geom)
  int sampled;                       // 0: every pixel that feeds an ommatidium; kEyeRays: that many rays per ommatidium (see nmf_eye_render)
  unsigned char rgb[5 + kMaxSpheres][4];   // materials: 0 sky, 1 ground A, 2 ground B, 3 terrain side wall, 4 own body, 5.. spheres
};

// constant-height cell of h(x, y) that holds (x, y): bounds (+-inf where unbounded) and level — the same arithmetic as
// the physics' terrain_height (nmf_step.hip) and as oracle/sensors_oracle.py::terrain_cell
struct TerrainCell { float x0, x1, y0, y1, h; };
__device__ __forceinline__ TerrainCell cell_gapped(float block, float gap, float depth, float x) {
  const float period = block + gap;
  const float base = floorf(x / period) * period;
  const bool on = x - base < block;
  return TerrainCell{on ? base : base + block, on ? base + block : base + period, -INFINITY, INFINITY, on ? 0.f : -depth};
}
__device__ __forceinline__ TerrainCell cell_blocks(float size, float height, float x, float y) {
  const float i = floorf(x / size), j = floorf(y / size);
  const float sum = i + j;
  const float par = sum - 2.f * floorf(sum / 2.f);
  return TerrainCell{i * size, (i + 1.f) * size, j * size, (j + 1.f) * size, par != 0.f ? height : 0.f};
}
__device__ __forceinline__ TerrainCell terrain_cell(int kind, const float* p, float x, float y) {
  if (kind == 1) return cell_gapped(p[0], p[1], p[2], x);
  if (kind == 2) return cell_blocks(p[0], p[1], x, y);
  if (kind == 3) {
    const float st = floorf(x / p[3]);
    const float k = st - 3.f * floorf(st / 3.f);
    const float s0 = st * p[3], s1 = (st + 1.f) * p[3];
    if (k == 1.f) { TerrainCell c = cell_gapped(1.0f, p[1], p[2], x); c.x0 = fmaxf(c.x0, s0); c.x1 = fminf(c.x1, s1); return c; }
    if (k == 2.f) { TerrainCell c = cell_blocks(p[0], 0.35f, x, y); c.x0 = fmaxf(c.x0, s0); c.x1 = fminf(c.x1, s1); return c; }
    return TerrainCell{s0, s1, -INFINITY, INFINITY, 0.f};
  }
  return TerrainCell{-INFINITY, INFINITY, -INFINITY, INFINITY, 0.f};
}

// camera-frame ray of a raw pixel (equidistant fisheye).  (A per-pixel ray table shared by all worlds was tried: 16 B
// per ray through L2 is slower than recomputing the lens model, 5.5 vs 2.4 ms per 8192 eye views.)
__device__ __forceinline__ V3 eye_ray(int row, int col, float cx, float cy, float inv_half_h, float half_fov, float& theta) {
  const float u = ((float)col + 0.5f - cx) * inv_half_h, v = ((float)row + 0.5f - cy) * inv_half_h;
  const float rho2 = u * u + v * v;
  const float rinv = __builtin_amdgcn_rsqf(fmaxf(rho2, 1e-12f));
  theta = rho2 * rinv * half_fov;
  const float st = __sinf(theta), ct = __cosf(theta);
  return v3(st * u * rinv, -st * v * rinv, -ct);          // x right, y up, looks along -z
}

// floor(x) as an integer in one instruction (saturating, NaN -> 0: no guard needed for rays that miss)
__device__ __forceinline__ int floor_int(float x) {
  int r;
  asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(r) : "v"(x));
  return r;
}

// sin(theta) / theta and cos(theta) as polynomials in x = theta^2 for theta <= 2.1 (a lens up to 240 degrees): near-minimax fits of
// degree 4 and 5 (Chebyshev nodes over [0, 4.41]; 7.6e-8 and 6.9e-9 from the functions in exact arithmetic, 2-3e-7 evaluated in
// float32 — the rounding of the evaluation itself; the Taylor polynomials of degree 7 and 8 they replace were no closer in
// float32 and cost six more multiply-adds per ray).  One definition: the pixel-exact and the sampled kernel must agree bit for bit.
__device__ __forceinline__ void lens_poly(float x, float& sc, float& cs) {
  sc = 2.4920499623e-06f;
  sc = fmaf(sc, x, -1.9741108406e-04f); sc = fmaf(sc, x, 8.3317681782e-03f); sc = fmaf(sc, x, -1.6666580795e-01f); sc = fmaf(sc, x, 9.9999992450e-01f);
  cs = -2.4918686236e-07f;
  cs = fmaf(cs, x, 2.4672689480e-05f); cs = fmaf(cs, x, -1.3885964226e-03f); cs = fmaf(cs, x, 4.1666365781e-02f); cs = fmaf(cs, x, -4.9999988662e-01f);
  cs = fmaf(cs, x, 9.9999999307e-01f);
}

// Waves per SIMD the register allocation aims at (measured per 8192 views with the own body: 5 waves 2.88 ms, 6 waves
// 2.71, 7 waves 2.55 — with 24 spilled registers, still the fastest —, 8 waves 3.00): latency hiding beats the spills until
// the allocation drops to 64 registers.
#ifndef NMF_EYE_WAVES
#define NMF_EYE_WAVES 7
#endif
// SAMPLED: the sampled mode (nmf_eye_params::rays_per_ommatidium = 16) as an instantiation of its own — the pixel-exact kernel
// keeps its register allocation, this one has few live values and takes eight waves per SIMD.
// RELIEF: the world has a terrain (flygym_amd/compose/world.py kinds 1-3) — the cell-by-cell walk of a ray through it exists in
// these instantiations only; flat worlds run without its code and registers.
// FRAMES: the call also wants the raw frames (inspection, parity tests); the readings-only instantiations carry neither the frame
// bytes' registers nor their code through the pixel loop.
template <bool SAMPLED, bool RELIEF, bool FRAMES>
__global__ void __launch_bounds__(kEyeThreads) __attribute__((amdgpu_waves_per_eu(SAMPLED ? 8 : NMF_EYE_WAVES, SAMPLED ? 8 : NMF_EYE_WAVES)))
nmf_eye_kernel(EyeArgs A, const float* __restrict__ seg_xpos, const float* __restrict__ seg_xquat, int nseg,
               const float* __restrict__ spheres, const int* __restrict__ cap_seg, const float* __restrict__ cap_geom,
               const u32x4* __restrict__ plan, const int* __restrict__ visit, const float* __restrict__ cones, const float4* __restrict__ chunk_cones, int n_groups,
               const int16_t* __restrict__ id_map, const int* __restrict__ slot_omm,
               const uint8_t* __restrict__ pale, const float* __restrict__ inv_norm, int n_omm,
               uint8_t* __restrict__ frames_out, float* __restrict__ omm_out) {
  __shared__ unsigned int acc[kMaxOmmatidia];
  __shared__ float sph[kMaxSpheres][4];
  __shared__ unsigned int mats[6 + kMaxSpheres];        // [0] black (outside the fisheye), [1 + m] material m
  // body capsules relative to the camera, world axes: 0-2 pa, 3-5 ba = pb - pa, 6 ba.ba, 7 -ba.pa, 8 quadratic's constant,
  // 9-11 pb, 12 pa.pa - r^2, 13 pb.pb - r^2, 14-16 unit direction to the bounding sphere's centre, 17 / 18 cos / sin of its
  // angular radius (+ margin)
  __shared__ float capd[kMaxCaps][20];
  // a capsule's angular cover: three discs (unit direction, cos / sin of the angular radius) over its thirds
  __shared__ float capc[kMaxCaps][3][5];
  __shared__ float sphc[kMaxSpheres][8];     // unit direction to the sphere's centre, cos / sin of its angular radius (+ margin)
  const int w = blockIdx.x >> 1, eye = blockIdx.x & 1;
  for (int i = threadIdx.x; i < n_omm; i += kEyeThreads) acc[i] = 0u;
  if (threadIdx.x < A.n_spheres * 4) (&sph[0][0])[threadIdx.x] = spheres[(size_t)w * A.sphere_stride + threadIdx.x];
  if (threadIdx.x < 6 + kMaxSpheres)
    mats[threadIdx.x] = threadIdx.x == 0 ? 0u : *reinterpret_cast<const unsigned int*>(A.rgb[threadIdx.x - 1]);
  __syncthreads();
  // the common materials' rgb words, wave-uniform: scalar registers
  auto rgb_word = [&](int mtl) { unsigned int v; __builtin_memcpy(&v, A.rgb[mtl], 4); return v; };      // (kernel argument: a scalar load)
  const unsigned int w_sky = rgb_word(0), w_ga = rgb_word(1), w_gb = rgb_word(2), w_wall = rgb_word(3), w_body = rgb_word(4), w_gx = w_ga ^ w_gb;
  // camera pose (uniform over the workgroup)
  const int sg = A.eye_seg[eye];
  float Rs[9];
  qmat(Rs, ldq(seg_xquat + ((size_t)w * nseg + sg) * 4));
  const V3 cam_v = ld3(seg_xpos + ((size_t)w * nseg + sg) * 3) + mat_vec(Rs, ld3(A.rel_pos[eye]));
  float R[9];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j)
      R[3 * i + j] = Rs[3 * i] * A.rel_mat[eye][j] + Rs[3 * i + 1] * A.rel_mat[eye][3 + j] + Rs[3 * i + 2] * A.rel_mat[eye][6 + j];
  // the view's constants are the same in every lane: scalar registers (they took 16 of the pixel loop's 72 vector registers,
  // and 24 more values went to scratch)
  auto uni = [](float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v))); };
#pragma unroll
  for (int i = 0; i < 9; ++i) R[i] = uni(R[i]);
  const V3 cam = v3(uni(cam_v.x), uni(cam_v.y), uni(cam_v.z));
  const int n_pix = A.height * A.width;
  const float inv_half_h = 2.0f / (float)A.height, cx = 0.5f * (float)A.width, cy = 0.5f * (float)A.height;
  const float inv_cs = 1.0f / A.checker_size;
  const float hz = cam.z - A.ground_z;
  // per sphere: (camera - centre, |camera - centre|^2 - r^2), uniform over the workgroup
  V3 oc = v3(0.f, 0.f, 0.f); float cc = 0.f;
  if (threadIdx.x < A.n_spheres) {
    const int s = threadIdx.x;
    oc = cam - v3(sph[s][0], sph[s][1], sph[s][2]);
    cc = dot(oc, oc) - sph[s][3] * sph[s][3];
  }
  __syncthreads();
  if (threadIdx.x < A.n_spheres) {
    float* q = sph[threadIdx.x];
    const float r = q[3];
    q[0] = oc.x; q[1] = oc.y; q[2] = oc.z; q[3] = cc;
    const float dist = sqrtf(dot(oc, oc));
    const float inv = dist > 1e-6f ? 1.0f / dist : 0.f;
    float* c = sphc[threadIdx.x];
    c[0] = -oc.x * inv; c[1] = -oc.y * inv; c[2] = -oc.z * inv;
    const float ang = dist > r ? asinf(r / dist) + 0.03f : 3.2f;
    c[3] = ang < 3.1f ? cosf(ang) : -2.f; c[4] = ang < 3.1f ? sinf(ang) : 0.f;
  }
  if (threadIdx.x < A.n_caps) {
    const int c = threadIdx.x, cs = cap_seg[c];
    const float* g = cap_geom + 7 * c;
    float Rc[9];
    qmat(Rc, ldq(seg_xquat + ((size_t)w * nseg + cs) * 4));
    const V3 xp = ld3(seg_xpos + ((size_t)w * nseg + cs) * 3);
    const V3 pa = (xp + mat_vec(Rc, ld3(g))) - cam, pb = (xp + mat_vec(Rc, ld3(g + 3))) - cam;
    const float r = g[6];
    const V3 ba = pb - pa;
    const float baba = dot(ba, ba), baoa = -dot(ba, pa), oaoa = dot(pa, pa);
    float* q = capd[c];
    q[0] = pa.x; q[1] = pa.y; q[2] = pa.z; q[3] = ba.x; q[4] = ba.y; q[5] = ba.z; q[6] = baba; q[7] = baoa;
    q[8] = baba * oaoa - baoa * baoa - r * r * baba;
    q[9] = pb.x; q[10] = pb.y; q[11] = pb.z; q[12] = oaoa - r * r; q[13] = dot(pb, pb) - r * r;
    q[14] = 0.f; q[15] = 0.f; q[16] = 0.f; q[17] = 0.f; q[18] = 0.f; q[19] = 0.f;
    // angular cover: the thirds of the axis, each inside a sphere of radius (length / 6 + r) — a leg segment's single
    // bounding sphere covers 3-10 x its angular area
    const float Rb = sqrtf(baba) * (1.0f / 6.0f) + r;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const V3 mid = pa + ((2.f * (float)i + 1.f) * (1.0f / 6.0f)) * ba;
      const float dist = sqrtf(dot(mid, mid));
      const float inv = dist > 1e-6f ? 1.0f / dist : 0.f;
      const float ang = dist > Rb ? asinf(Rb / dist) + 0.004f : 3.2f;       // camera inside the bound: always a candidate
      float* cq = capc[c][i];
      cq[0] = mid.x * inv; cq[1] = mid.y * inv; cq[2] = mid.z * inv;
      cq[3] = ang < 3.1f ? cosf(ang) : -2.f; cq[4] = ang < 3.1f ? sinf(ang) : 0.f;
    }
  }
  __syncthreads();
  uint8_t* const fout = FRAMES && frames_out ? frames_out + (size_t)blockIdx.x * n_pix * 3 : nullptr;
  // the visit plan lists the chunks to render (readings only: those that touch an ommatidium — 43 % of the frame lies outside
  // the lattice) in groups of 64, one group per wave and turn
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int grp = wave; grp < n_groups; grp += kEyeThreads / 64) {
    // (sampled mode: a group is kEyeSlots ommatidia = kEyeSlots / 4 turns of 64 rays under one culling; `ch` is the first turn's pixel)
    const int ch = visit[(SAMPLED ? grp * (kEyeSlots / 4) : grp) * 64 + lane];
    // the group's cone(s), world axes: one, or — groups of chunks that wrap to the next image row — one per piece
    const float* cn = cones + (size_t)__builtin_amdgcn_readfirstlane(grp) * 12;
    const bool two = cn[8] != 0.f;
    V3 gw[2]; float g_cos[2], g_sin[2];
#pragma unroll
    for (int pc = 0; pc < 2; ++pc) {
      const float ax = cn[4 * pc], ay = cn[4 * pc + 1], az = cn[4 * pc + 2];
      gw[pc] = v3(R[0] * ax + R[1] * ay + R[2] * az, R[3] * ax + R[4] * ay + R[5] * az, R[6] * ax + R[7] * ay + R[8] * az);
      g_cos[pc] = cn[4 * pc + 3]; g_sin[pc] = sqrtf(fmaxf(1.f - g_cos[pc] * g_cos[pc], 0.f));
    }
    // does a cone (axis a, half-angle by cos / sin) meet a disc of angular radius (qc, qs) about the unit direction q?
    auto meets = [](V3 a, float c_cos, float c_sin, float qx, float qy, float qz, float qc, float qs) {
      const float ca = qx * a.x + qy * a.y + qz * a.z;
      // (no short-circuit: three compares and two scalar ors cost less than the branches around them)
      return (int)(qc < -1.5f) | (int)(ca >= c_cos * qc - c_sin * qs) | (int)(c_sin * qc + c_cos * qs <= 0.f);      // (last: the two angles add up to pi or more)
    };
    auto cone_sees = [&](float qx, float qy, float qz, float qc, float qs) {
      int r = meets(gw[0], g_cos[0], g_sin[0], qx, qy, qz, qc, qs);
      if (two) r |= meets(gw[1], g_cos[1], g_sin[1], qx, qy, qz, qc, qs);         // (wave-uniform)
      return r;
    };
    int sees_cap = 0;
    if (lane < A.n_caps) {
#pragma unroll
      for (int i = 0; i < 3; ++i) { const float* cq = capc[lane][i]; sees_cap |= cone_sees(cq[0], cq[1], cq[2], cq[3], cq[4]); }
    }
    const unsigned long long grp_caps = __ballot(sees_cap != 0);
    const unsigned int grp_sph = (unsigned int)__ballot(lane < A.n_spheres && 0 != cone_sees(sphc[lane < A.n_spheres ? lane : 0][0], sphc[lane < A.n_spheres ? lane : 0][1],
                                                                                      sphc[lane < A.n_spheres ? lane : 0][2], sphc[lane < A.n_spheres ? lane : 0][3], sphc[lane < A.n_spheres ? lane : 0][4]));
    // some ray of the group points below the horizon (a cone wider than a hemisphere always does: sin falls again beyond 90 degrees)
    const bool grp_ground = g_cos[0] <= 0.f || gw[0].z < g_sin[0] + 1e-3f || (two && (g_cos[1] <= 0.f || gw[1].z < g_sin[1] + 1e-3f));
    constexpr bool sampled = SAMPLED;         // the visit list holds PIXELS, kEyeRays per ommatidium, one ray per lane
    if (!sampled && ch < 0) continue;
    // A group whose cone meets nothing — no ray below the horizon, no sphere, no capsule — sees the sky in every pixel: no ray
    // is built at all (the upper half of an eye's image, less what the fly's own body covers).  Same bytes as the pixel loop
    // would produce (the chunks it applies to are those of the polynomial path: never behind the lens' full sphere).
    const bool sky_only = !grp_ground && grp_sph == 0u && grp_caps == 0ull && cn[9] != 0.f;
#ifdef NMF_EYE_STATS
    if (lane == 0) atomicAdd(&g_eye_stats[7], (sky_only ? 1ull : 0ull) | ((!grp_ground && grp_sph == 0u) ? 1ull << 32 : 0ull));
#endif
    const int chunk = sampled ? (ch < 0 ? 0 : (ch & 0xffffff) >> 4) : ch;      // (sampled: a visit entry is pixel | flags)
    const u32x4 pl = plan[chunk];
    const bool planned = !(pl.y & 0x10000u);
    int row = (chunk * 16) / A.width, col = chunk * 16 - row * A.width;
    unsigned int tot = 0u, sA = 0u, sAB = 0u;
    // body capsules this chunk can see: its rays lie in a cone about the mean of its first and last ray — two cones for
    // a chunk that wraps to the next image row (one per row piece).  A capsule is a candidate if its bounding sphere's
    // angular disc reaches into a cone.
    unsigned long long cand = 0ull, ucand = 0ull;         // this chunk's candidates / the union over the wave's chunks
    if (sampled) { cand = grp_caps; ucand = grp_caps; }   // single rays: the group's candidates as they are
    else if (grp_caps) {
      V3 cw[2]; float c_cos[2], c_sin[2];
#pragma unroll
      for (int pc = 0; pc < 2; ++pc) {
        if (pc == 1 && !two) break;
        const float4 cc = chunk_cones[(grp * 64 + lane) * 2 + pc];          // bounding cone of the piece's rays, camera frame
        cw[pc] = v3(R[0] * cc.x + R[1] * cc.y + R[2] * cc.z, R[3] * cc.x + R[4] * cc.y + R[5] * cc.z, R[6] * cc.x + R[7] * cc.y + R[8] * cc.z);
        c_cos[pc] = cc.w; c_sin[pc] = sqrtf(fmaxf(1.f - cc.w * cc.w, 0.f));
      }
      for (unsigned long long gm = grp_caps; gm; gm &= gm - 1ull) {
        const int c = __ffsll((long long)gm) - 1;
        int hit = 0;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          const float* cq = capc[c][i];
          hit |= meets(cw[0], c_cos[0], c_sin[0], cq[0], cq[1], cq[2], cq[3], cq[4]);
          if (two) hit |= meets(cw[1], c_cos[1], c_sin[1], cq[0], cq[1], cq[2], cq[3], cq[4]);
        }
        if (hit) cand |= 1ull << c;
        if (__any(hit != 0)) ucand |= 1ull << c;
      }
    }
#ifdef NMF_EYE_STATS
    if (lane == 0) { atomicAdd(&g_eye_stats[0], 1ull); atomicAdd(&g_eye_stats[1], (unsigned long long)__popcll(grp_caps)); atomicAdd(&g_eye_stats[2], (unsigned long long)__popcll(ucand));
                     atomicAdd(&g_eye_stats[4], grp_ground ? 1ull : 0ull); atomicAdd(&g_eye_stats[5], (unsigned long long)__popc(grp_sph)); atomicAdd(&g_eye_stats[6], ucand ? 1ull : 0ull); }
    atomicAdd(&g_eye_stats[3], (unsigned long long)__popcll(cand));
#endif
    // what a world-axis ray sees: ground plane (checker) where the group's cone reaches below the horizon, then the spheres
    // and capsules the group can see; the nearest positive hit wins
    // (returns the material's rgb word: the sky's and the ground's — nearly every ray's — are in scalar registers; black = 0)
    // (objs: the group sees a sphere or a capsule at all — wave-uniform, a scalar branch per ray; without one the ground's distance
    // is never compared with anything)
    const bool objs = __builtin_amdgcn_readfirstlane((grp_sph != 0u || ucand != 0ull || (RELIEF && A.terrain_kind != 0)) ? 1 : 0) != 0;
    auto scene = [&](const V3 d) {
      unsigned int rgb = w_sky;
      float tbest = INFINITY;
      if (grp_ground) {
        const float t = -hz * __builtin_amdgcn_rcpf(d.z);
        const bool ghit = d.z < 0.f && t > 0.f;
        const float gx = (cam.x + t * d.x) * inv_cs, gy = (cam.y + t * d.y) * inv_cs;
        // checker parity as a mask (bit 0 of the cell sum, sign-extended): ground A xor (mask and (A xor B)) — no select between two scalars
        const unsigned int pm = (unsigned int)(((floor_int(gx) + floor_int(gy)) << 31) >> 31);
        rgb = ghit ? (w_ga ^ (pm & w_gx)) : w_sky;
        if (objs) tbest = ghit ? t : INFINITY;
      }
      if (!objs) return rgb;
      if constexpr (RELIEF) if (grp_ground && A.terrain_kind != 0 && d.z < 0.f) {
        // relief: follow the ray through the cells of h(x, y) from the highest level down; a cell entered below its level
        // is a side wall, else its top is hit if the ray reaches the level before leaving the cell
        const float rdz = 1.0f / d.z;
        float tc = fmaxf(0.f, (A.ground_z + A.terrain[4] - cam.z) * rdz);
#pragma unroll 1
        for (int it = 0; it < kMaxTerrainCells; ++it) {
          const float tp = tc + kTerrainEps;
          const TerrainCell c = terrain_cell(A.terrain_kind, A.terrain, cam.x + tp * d.x, cam.y + tp * d.y);
          const float h = c.h + A.ground_z;
          const float z_in = cam.z + tc * d.z;
          if (z_in < h - kTerrainWallTol) { tbest = tc; rgb = w_wall; break; }
          const float tx = d.x > 0.f ? (c.x1 - cam.x) / d.x : (d.x < 0.f ? (c.x0 - cam.x) / d.x : INFINITY);
          const float ty = d.y > 0.f ? (c.y1 - cam.y) / d.y : (d.y < 0.f ? (c.y0 - cam.y) / d.y : INFINITY);
          const float t_out = fminf(tx, ty);
          const float t_h = (h - cam.z) * rdz;
          if (t_h <= t_out) {
            const float qx = (cam.x + t_h * d.x) * inv_cs, qy = (cam.y + t_h * d.y) * inv_cs;
            tbest = t_h; rgb = ((floor_int(qx) + floor_int(qy)) & 1) ? w_gb : w_ga;
            break;
          }
          tc = t_out;
        }
      }
      for (unsigned int sm = grp_sph; sm; sm &= sm - 1u) {
        const int s = __ffs((int)sm) - 1;
        const float b = sph[s][0] * d.x + sph[s][1] * d.y + sph[s][2] * d.z;
        const float disc = b * b - sph[s][3];
        const float ts = -b - __builtin_amdgcn_sqrtf(fmaxf(disc, 0.f));
        const bool ok = disc > 0.f && ts > 0.f && ts < tbest;
        tbest = ok ? ts : tbest; rgb = ok ? mats[6 + s] : rgb;
      }
      for (unsigned long long cm = ucand; cm; cm &= cm - 1ull) {
        const int ci = __ffsll((long long)cm) - 1;
        if (!((cand >> ci) & 1ull)) continue;
        const float* q = capd[ci];
        const float bard = q[3] * d.x + q[4] * d.y + q[5] * d.z;
        const float rdoa = -(q[0] * d.x + q[1] * d.y + q[2] * d.z);
        const float a = q[6] - bard * bard;
        const float b = q[6] * rdoa - q[7] * bard;
        const float hq = b * b - a * q[8];
        const float tb = (-b - sqrtf(fmaxf(hq, 0.f))) / a;
        const float yy = q[7] + tb * bard;
        const bool cyl = hq >= 0.f && a > 1e-12f;
        float tcap = INFINITY;
        if (cyl && yy > 0.f && yy < q[6] && tb > 0.f) tcap = tb;
        else {
          const bool use_a = !(yy > 0.f) || !cyl;
          const float bb = use_a ? rdoa : -(q[9] * d.x + q[10] * d.y + q[11] * d.z);
          const float hh = bb * bb - (use_a ? q[12] : q[13]);
          const float te = -bb - sqrtf(fmaxf(hh, 0.f));
          if (hh > 0.f && te > 0.f) tcap = te;
        }
        if (tcap < tbest) { tbest = tcap; rgb = w_body; }
      }
      return rgb;
    };
    if constexpr (SAMPLED) {
      // Sampled mode: a lane's ray is a pixel of the raw frame — computed exactly as the pixel-exact mode computes that pixel
      // (same lens branch for its chunk: the polynomials inside a row, eye_ray for a chunk that wraps or is unplanned), so its
      // colour is the frame's.  The kEyeRays lanes of an ommatidium form one DPP row: their chosen colour bytes are summed
      // across the row and the row's last lane adds them to the ommatidium.
      const float hf2 = A.half_fov * A.half_fov;
      // (the plan's lists carry the pixel's path and the ommatidium's pale flag with them, and the next turn's two words are
      // requested while this turn's rays are cast: one memory round trip per turn behind arithmetic instead of three in front of it)
      int pxw = ch;
      int slw = (grp * (kEyeSlots / 4)) * 4 + (lane >> 4) < n_omm ? slot_omm[(grp * (kEyeSlots / 4)) * 4 + (lane >> 4)] : 0;
#pragma unroll 1
      for (int turn = 0; turn < kEyeSlots / 4; ++turn) {
        const int slot = (grp * (kEyeSlots / 4) + turn) * 4 + (lane >> 4);
        const int pxw_cur = pxw, slw_cur = slw;
        if (grp_caps) {      // the capsules this TURN's 64 rays can meet (wave-uniform: lane = capsule, as for the group)
          const float4 tcn = chunk_cones[__builtin_amdgcn_readfirstlane(grp * (kEyeSlots / 4) + turn)];
          const V3 tw = v3(R[0] * tcn.x + R[1] * tcn.y + R[2] * tcn.z, R[3] * tcn.x + R[4] * tcn.y + R[5] * tcn.z, R[6] * tcn.x + R[7] * tcn.y + R[8] * tcn.z);
          const float t_cos = tcn.w, t_sin = sqrtf(fmaxf(1.f - tcn.w * tcn.w, 0.f));
          int sees = 0;
          if (lane < A.n_caps) {
#pragma unroll
            for (int i = 0; i < 3; ++i) { const float* cq = capc[lane][i]; sees |= meets(tw, t_cos, t_sin, cq[0], cq[1], cq[2], cq[3], cq[4]); }
          }
          cand = __ballot(sees != 0) & grp_caps; ucand = cand;
        }
        if (turn + 1 < kEyeSlots / 4) {
          pxw = visit[(grp * (kEyeSlots / 4) + turn + 1) * 64 + lane];
          slw = slot + 4 < n_omm ? slot_omm[slot + 4] : 0;
        }
        const int px = pxw_cur < 0 ? -1 : (pxw_cur & 0xffffff);
        const bool poly = pxw_cur >= 0 && ((pxw_cur >> 30) & 1) != 0;       // inside one image row, run plan: the polynomial path
        const int omm = slw_cur & 0xffff;
        const bool is_pale = ((slw_cur >> 30) & 1) != 0;
        unsigned int val = 0u;
        if (px >= 0 && sky_only && poly) {
          const unsigned int rgbw = w_sky;            // (a pixel of the polynomial path in a group that sees the sky only)
          val = is_pale ? ((rgbw >> 16) & 0xffu) : ((rgbw >> 8) & 0xffu);
        } else if (px >= 0) {
          const int prow = px / A.width, pcol = px - prow * A.width;
          unsigned int rgbw;
          if (poly && cn[9] != 0.f) {
            const float vv = ((float)prow + 0.5f - cy) * inv_half_h;
            const float v2 = vv * vv;
            const V3 rowv = v3(-vv * R[1], -vv * R[4], -vv * R[7]);
            const float u = ((float)pcol + 0.5f - cx) * inv_half_h;
            const float x = fmaf(u, u, v2) * hf2;
            float sc, cs;
            lens_poly(x, sc, cs);
            const float s1 = sc * A.half_fov;
            const V3 d = v3(fmaf(s1, fmaf(u, R[0], rowv.x), -cs * R[2]), fmaf(s1, fmaf(u, R[3], rowv.y), -cs * R[5]), fmaf(s1, fmaf(u, R[6], rowv.z), -cs * R[8]));
            rgbw = scene(d);
          } else {
            float theta;
            const V3 dc = eye_ray(prow, pcol, cx, cy, inv_half_h, A.half_fov, theta);
            const float dx = dc.x, dy = dc.y, dz = dc.z;
            const V3 d = v3(R[0] * dx + R[1] * dy + R[2] * dz, R[3] * dx + R[4] * dy + R[5] * dz, R[6] * dx + R[7] * dy + R[8] * dz);
            rgbw = scene(d);
            rgbw = theta > 3.14159265f ? 0u : rgbw;
          }
          val = is_pale ? ((rgbw >> 16) & 0xffu) : ((rgbw >> 8) & 0xffu);
        }
        // sum over the 16 lanes of the row (row_shr 8, 4, 2, 1 with zero fill: the row's last lane ends up with the total)
        int v = (int)val;
        v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);
        v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);
        v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);
        v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);
        if ((lane & 15) == 15 && slot < n_omm) atomicAdd(&acc[omm], (unsigned int)v);
      }
      continue;
    }
    if (!SAMPLED && sky_only && !two) {
      // (wave-uniform: a chunk with more than three runs — one in fifty, i.e. most waves hold one — adds its sixteen equal pixels
      // through the id map instead of sending the whole wave through the pixel loop)
      const unsigned int sky = mats[1];
      const unsigned int skyG = (sky >> 8) & 0xffu, skyB = (sky >> 16) & 0xffu;
      const unsigned int pm = pl.z & 0xffffu, mA = pl.w & 0xffffu, mAB = pl.w >> 16;
      if (planned) {
        tot = (unsigned int)__popc(pm) * skyB + (unsigned int)__popc(~pm & 0xffffu) * skyG;
        sA = (unsigned int)__popc(pm & mA) * skyB + (unsigned int)__popc(~pm & mA) * skyG;
        sAB = (unsigned int)__popc(pm & mAB) * skyB + (unsigned int)__popc(~pm & mAB) * skyG;
      } else {
#pragma unroll 1
        for (int k = 0; k < 16; ++k) {
          const int id = (int)((unsigned short)id_map[(size_t)ch * 16 + k] & 0x7fffu);
          if (id > 0) atomicAdd(&acc[id - 1], ((pm >> k) & 1u) ? skyB : skyG);
        }
      }
      if (FRAMES && fout) {
        const unsigned int r0 = sky & 0xffu, r1 = skyG, r2 = skyB;
        const unsigned int w0 = r0 | (r1 << 8) | (r2 << 16) | (r0 << 24), w1 = r1 | (r2 << 8) | (r0 << 16) | (r1 << 24), w2 = r2 | (r0 << 8) | (r1 << 16) | (r2 << 24);
        unsigned int* o = reinterpret_cast<unsigned int*>(fout + (size_t)ch * 48);
#pragma unroll
        for (int g = 0; g < 4; ++g) { o[3 * g] = w0; o[3 * g + 1] = w1; o[3 * g + 2] = w2; }
      }
    } else if (!two && planned && cn[9] != 0.f) {
      // The common case — a chunk inside one image row, three or fewer runs, the whole image within the lens polynomials' range
      // (cn[9]; both are properties of the chunk / the lens, so a pixel takes the same path whatever the call renders) — on a diet: the row's part of the ray is hoisted, sin(theta) / rho and cos(theta) are polynomials in
      // theta^2 (lens_poly: no rsq / sin / cos), the chosen colour byte
      // of a pixel goes into packed words with one v_perm each and the run sums are v_dot4 against the plan's masks.
      const float vv = ((float)row + 0.5f - cy) * inv_half_h;
      const float hf2 = A.half_fov * A.half_fov, v2 = vv * vv;
      const V3 rowv = v3(-vv * R[1], -vv * R[4], -vv * R[7]);
      // (four pixels per turn of a rolled loop: the scene code exists four times, not sixteen — the kernel has to fit the
      // instruction cache too)
      auto chunk_pixels = [&](auto&& colour_of) {      // colour_of(ray) -> the material's rgb word
        // ((float)(col + k) + 0.5 - cx = ((float)col + 0.5 - cx) + k exactly: multiples of 0.5 far below 2^24)
        float uc = (float)col + 0.5f - cx;
#pragma unroll 1
        for (int g = 0; g < 4; ++g, uc += 4.f) {
          unsigned int G4 = 0u, B4 = 0u, raw[3] = {0u, 0u, 0u};
#pragma unroll
          for (int k4 = 0; k4 < 4; ++k4) {
            const float u = (uc + (float)k4) * inv_half_h;
            const float x = fmaf(u, u, v2) * hf2;               // theta^2
            float sc, cs;                                       // sin(theta) / theta, cos(theta)
            lens_poly(x, sc, cs);
            const float s1 = sc * A.half_fov;                   // sin(theta) / rho
            const V3 d = v3(fmaf(s1, fmaf(u, R[0], rowv.x), -cs * R[2]), fmaf(s1, fmaf(u, R[3], rowv.y), -cs * R[5]), fmaf(s1, fmaf(u, R[6], rowv.z), -cs * R[8]));
            const unsigned int rgbw = colour_of(d);
            constexpr unsigned int keep = 0x03020100u;
            G4 = __builtin_amdgcn_perm(rgbw, G4, (keep & ~(0xffu << (8 * k4))) | (5u << (8 * k4)));
            B4 = __builtin_amdgcn_perm(rgbw, B4, (keep & ~(0xffu << (8 * k4))) | (6u << (8 * k4)));
            if (FRAMES && fout) {
#pragma unroll
              for (int cidx = 0; cidx < 3; ++cidx) {
                const int bpos = 3 * k4 + cidx;
                raw[bpos >> 2] |= ((rgbw >> (8 * cidx)) & 0xffu) << ((bpos & 3) * 8);
              }
            }
          }
          const unsigned int P4 = ((((pl.z >> (4 * g)) & 0xfu) * 0x00204081u) & 0x01010101u) * 0xffu;
          const unsigned int V4 = (B4 & P4) | (G4 & ~P4);
          tot = __builtin_amdgcn_udot4(V4, 0x01010101u, tot, false);
          sA = __builtin_amdgcn_udot4(V4, (((pl.w >> (4 * g)) & 0xfu) * 0x00204081u) & 0x01010101u, sA, false);
          sAB = __builtin_amdgcn_udot4(V4, (((pl.w >> (16 + 4 * g)) & 0xfu) * 0x00204081u) & 0x01010101u, sAB, false);
          if (FRAMES && fout) {
            unsigned int* o = reinterpret_cast<unsigned int*>(fout + (size_t)ch * 48) + 3 * g;
            o[0] = raw[0]; o[1] = raw[1]; o[2] = raw[2];
          }
        }
      };
      // (Tried, round 5: a straight-line loop for the groups that can only see the checker or the sky — more than half of them —
      // with the three colours in registers: 49 spilled registers at this occupancy, 2.96 ms against 2.43, and bytes that differ
      // from scene()'s at checker edges, which the sampled mode's bit-equality with these frames does not allow.)
      chunk_pixels([&](const V3 d) { return scene(d); });
    } else {
      // everything else (a chunk that wraps to the next row, more than three runs, a lens beyond the polynomials' range): per
      // pixel, rolled
#pragma unroll 1
      for (int k = 0; k < 16; ++k) {
        float theta;
        const V3 dc = eye_ray(row, col, cx, cy, inv_half_h, A.half_fov, theta);
        const float dx = dc.x, dy = dc.y, dz = dc.z;
        const V3 d = v3(R[0] * dx + R[1] * dy + R[2] * dz, R[3] * dx + R[4] * dy + R[5] * dz, R[6] * dx + R[7] * dy + R[8] * dz);
        unsigned int rgbw = scene(d);
        rgbw = theta > 3.14159265f ? 0u : rgbw;                                // behind the fisheye's full sphere: black
        const unsigned int val = ((pl.z >> k) & 1u) ? ((rgbw >> 16) & 0xffu) : ((rgbw >> 8) & 0xffu);
        if (planned) {
          tot += val;
          sA += ((pl.w >> k) & 1u) ? val : 0u;
          sAB += ((pl.w >> (16 + k)) & 1u) ? val : 0u;
        } else {
          const unsigned int tid = (unsigned short)id_map[(size_t)ch * 16 + k];
          const int id = (int)(tid & 0x7fffu);
          if (id > 0) atomicAdd(&acc[id - 1], val);
        }
        if (FRAMES && fout) {
          uint8_t* o = fout + (size_t)ch * 48 + 3 * k;
          o[0] = (uint8_t)(rgbw & 0xffu); o[1] = (uint8_t)((rgbw >> 8) & 0xffu); o[2] = (uint8_t)((rgbw >> 16) & 0xffu);
        }
        if (++col == A.width) { col = 0; ++row; }
      }
    }
    if (planned) {
      const int ia = (int)(pl.x & 0x7fffu), ib = (int)((pl.x >> 16) & 0x7fffu), ic = (int)(pl.y & 0x7fffu);
      if (ia > 0) atomicAdd(&acc[ia - 1], sA);
      if (ib > 0) atomicAdd(&acc[ib - 1], sAB - sA);
      if (ic > 0) atomicAdd(&acc[ic - 1], tot - sAB);
    }
  }
  __syncthreads();
  if (omm_out) {
    float* dst = omm_out + (size_t)blockIdx.x * n_omm * 2;
    for (int i = threadIdx.x; i < n_omm; i += kEyeThreads) {
      const float vv = (float)acc[i] * (SAMPLED ? 1.0f / (255.0f * (float)kEyeRays) : inv_norm[i]);
      const bool pp = pale[i] != 0;
      dst[2 * i] = pp ? 0.f : vv;
      dst[2 * i + 1] = pp ? vv : 0.f;
    }
  }
}

}  // namespace nmf
