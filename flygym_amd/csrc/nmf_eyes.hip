// nmf_eyes.hip — compound-eye renderer fused with the ommatidia resample (gfx950).  SURVEY §8 (f) rank 2.
//
// The reference renders eye images with MuJoCo's OpenGL renderer / MJWarp's batch renderer and flygym 1.x then
// resampled them to ommatidia; this snapshot has neither the eye cameras nor the resample (only the constants in
// src/flygym/assets/model/legacy/flygym1_config.yaml:141-173), so the camera model and the scene are build-defined
//  and pinned by the numpy oracle oracle/sensors_oracle.py::render_eye_frames.
//
// One workgroup per (world, eye).  A lane owns 16 consecutive raw pixels (the resample's chunk, same run plan):
// per pixel it builds the equidistant-fisheye ray, rotates it into the world, intersects the ground plane (checker
// texture) and the spheres, takes the green or blue byte of the hit material and adds it to the chunk's run sums;
// three integer LDS atomics per chunk.  The 1.4 MB raw frame per fly never exists unless the caller asks for it
// (frames_out, for inspection and for the parity tests): compute-bound instead of HBM-bound.
#include "nmf_device.h"

namespace nmf {

constexpr int kEyeThreads = 512;
constexpr int kMaxSpheres = 8;

struct EyeArgs {
  int height, width;
  float half_fov;          // polar angle (rad) of the ray through the middle of the top / bottom image edge
  int eye_seg[2];
  float rel_pos[2][3];
  float rel_mat[2][9];     // camera axes in the parent segment frame (columns: right, up, back)
  float checker_size, ground_z;
  int n_spheres, sphere_stride;      // floats between consecutive worlds in `spheres` (0: shared by all worlds)
  unsigned char rgb[3 + kMaxSpheres][4];   // materials: 0 sky, 1 ground A, 2 ground B, 3.. spheres
};

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

__global__ void __launch_bounds__(kEyeThreads)
nmf_eye_kernel(EyeArgs A, const float* __restrict__ seg_xpos, const float* __restrict__ seg_xquat, int nseg,
               const float* __restrict__ spheres, const u32x4* __restrict__ plan, const int* __restrict__ active,
               const int16_t* __restrict__ id_map,
               const uint8_t* __restrict__ pale, const float* __restrict__ inv_norm, int n_omm,
               uint8_t* __restrict__ frames_out, float* __restrict__ omm_out) {
  __shared__ unsigned int acc[kMaxOmmatidia];
  __shared__ float sph[kMaxSpheres][4];
  __shared__ unsigned int mats[4 + kMaxSpheres];        // [0] black (outside the fisheye), [1 + m] material m
  const int w = blockIdx.x >> 1, eye = blockIdx.x & 1;
  for (int i = threadIdx.x; i < n_omm; i += kEyeThreads) acc[i] = 0u;
  if (threadIdx.x < A.n_spheres * 4) (&sph[0][0])[threadIdx.x] = spheres[(size_t)w * A.sphere_stride + threadIdx.x];
  if (threadIdx.x < 4 + kMaxSpheres)
    mats[threadIdx.x] = threadIdx.x == 0 ? 0u : *reinterpret_cast<const unsigned int*>(A.rgb[threadIdx.x - 1]);
  __syncthreads();
  // camera pose (uniform over the workgroup)
  const int sg = A.eye_seg[eye];
  float Rs[9];
  qmat(Rs, ldq(seg_xquat + ((size_t)w * nseg + sg) * 4));
  const V3 cam = ld3(seg_xpos + ((size_t)w * nseg + sg) * 3) + mat_vec(Rs, ld3(A.rel_pos[eye]));
  float R[9];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j)
      R[3 * i + j] = Rs[3 * i] * A.rel_mat[eye][j] + Rs[3 * i + 1] * A.rel_mat[eye][3 + j] + Rs[3 * i + 2] * A.rel_mat[eye][6 + j];
  const int n_pix = A.height * A.width, n_chunk = n_pix / 16;
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
  if (threadIdx.x < A.n_spheres) { float* q = sph[threadIdx.x]; q[0] = oc.x; q[1] = oc.y; q[2] = oc.z; q[3] = cc; }
  __syncthreads();
  uint8_t* fout = frames_out ? frames_out + (size_t)blockIdx.x * n_pix * 3 : nullptr;
  // readings only: visit just the chunks that touch an ommatidium (43 % of the frame lies outside the lattice)
  const int n_visit = fout ? n_chunk : active[n_chunk];
  for (int it = threadIdx.x; it < n_visit; it += kEyeThreads) {
    const int ch = fout ? it : active[it];
    const u32x4 pl = plan[ch];
    const bool planned = !(pl.y & 0x10000u);
    int row = (ch * 16) / A.width, col = ch * 16 - row * A.width;
    unsigned int tot = 0u, sA = 0u, sAB = 0u;
    unsigned int obytes[12] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      float theta;
      const V3 dc = eye_ray(row, col, cx, cy, inv_half_h, A.half_fov, theta);
      const float dx = dc.x, dy = dc.y, dz = dc.z;
      const V3 d = v3(R[0] * dx + R[1] * dy + R[2] * dz, R[3] * dx + R[4] * dy + R[5] * dz, R[6] * dx + R[7] * dy + R[8] * dz);
      // branch-free: ground plane (checker), then the spheres; the nearest positive hit wins
      const float t = -hz * __builtin_amdgcn_rcpf(d.z);
      const bool ghit = d.z < 0.f && t > 0.f;
      const float gx = (cam.x + t * d.x) * inv_cs, gy = (cam.y + t * d.y) * inv_cs;
      const int par = ((int)floorf(ghit ? gx : 0.f) + (int)floorf(ghit ? gy : 0.f)) & 1;
      int mat = ghit ? 1 + par : 0;
      float tbest = ghit ? t : INFINITY;
#pragma unroll 1
      for (int s = 0; s < A.n_spheres; ++s) {
        const float b = sph[s][0] * d.x + sph[s][1] * d.y + sph[s][2] * d.z;
        const float disc = b * b - sph[s][3];
        const float ts = -b - __builtin_amdgcn_sqrtf(fmaxf(disc, 0.f));
        const bool ok = disc > 0.f && ts > 0.f && ts < tbest;
        tbest = ok ? ts : tbest; mat = ok ? 3 + s : mat;
      }
      mat = theta > 3.14159265f ? -1 : mat;                                  // behind the fisheye's full sphere: black
      const unsigned int rgbw = mats[mat + 1];
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
      if (fout) {
#pragma unroll
        for (int cidx = 0; cidx < 3; ++cidx) {
          const int bpos = 3 * k + cidx;
          obytes[bpos >> 2] |= ((rgbw >> (8 * cidx)) & 0xffu) << ((bpos & 3) * 8);
        }
      }
      if (++col == A.width) { col = 0; ++row; }
    }
    if (planned) {
      const int ia = (int)(pl.x & 0x7fffu), ib = (int)((pl.x >> 16) & 0x7fffu), ic = (int)(pl.y & 0x7fffu);
      if (ia > 0) atomicAdd(&acc[ia - 1], sA);
      if (ib > 0) atomicAdd(&acc[ib - 1], sAB - sA);
      if (ic > 0) atomicAdd(&acc[ic - 1], tot - sAB);
    }
    if (fout) {
      u32x4* o = reinterpret_cast<u32x4*>(fout + (size_t)ch * 48);
      o[0] = u32x4{obytes[0], obytes[1], obytes[2], obytes[3]};
      o[1] = u32x4{obytes[4], obytes[5], obytes[6], obytes[7]};
      o[2] = u32x4{obytes[8], obytes[9], obytes[10], obytes[11]};
    }
  }
  __syncthreads();
  if (omm_out) {
    float* dst = omm_out + (size_t)blockIdx.x * n_omm * 2;
    for (int i = threadIdx.x; i < n_omm; i += kEyeThreads) {
      const float vv = (float)acc[i] * inv_norm[i];
      const bool pp = pale[i] != 0;
      dst[2 * i] = pp ? 0.f : vv;
      dst[2 * i + 1] = pp ? vv : 0.f;
    }
  }
}

}  // namespace nmf
