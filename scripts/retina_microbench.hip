// Load-pattern ceiling for the retina resample (diagnostic): stream the frames with (A) the kernel's per-thread
// 48-byte chunks (three 16-byte loads at stride 48 across lanes) or (B) fully coalesced 16-byte loads, and fold
// everything into one word per thread.  usage: scripts/retina_microbench.py
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int THREADS, bool NT>
__global__ void __launch_bounds__(THREADS) stream_kernel(const uint8_t* __restrict__ images, size_t frame_bytes, unsigned int* __restrict__ out) {
  const u32x4* src = reinterpret_cast<const u32x4*>(images + (size_t)blockIdx.x * frame_bytes);
  const int n16 = (int)(frame_bytes / 16);
  unsigned int acc = 0u;
  auto ld = [&](const u32x4* p) { return NT ? __builtin_nontemporal_load(p) : *p; };
  if (MODE == 0) {          // per-thread 48-byte chunks
    const int n_chunk = n16 / 3;
    for (int ch = threadIdx.x; ch < n_chunk; ch += THREADS) {
      const u32x4 a = ld(src + 3 * ch), b = ld(src + 3 * ch + 1), c = ld(src + 3 * ch + 2);
      acc ^= a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y ^ b.z ^ b.w ^ c.x ^ c.y ^ c.z ^ c.w;
    }
  } else {                  // coalesced: lane l reads 16 B at 16 l, four loads in flight
    int i = threadIdx.x;
    for (; i + 3 * THREADS < n16; i += 4 * THREADS) {
      const u32x4 a = ld(src + i), b = ld(src + i + THREADS), c = ld(src + i + 2 * THREADS), d = ld(src + i + 3 * THREADS);
      acc ^= a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y ^ b.z ^ b.w ^ c.x ^ c.y ^ c.z ^ c.w ^ d.x ^ d.y ^ d.z ^ d.w;
    }
    for (; i < n16; i += THREADS) { const u32x4 a = ld(src + i); acc ^= a.x ^ a.y ^ a.z ^ a.w; }
  }
  if (acc == 0x12345678u) out[blockIdx.x] = acc;
}

extern "C" int retina_stream(const uint8_t* images, int n_images, size_t frame_bytes, unsigned int* out, int mode, int threads, int nt, void* stream) {
  hipStream_t s = (hipStream_t)stream;
#define L(M, T, N) hipLaunchKernelGGL((stream_kernel<M, T, N>), dim3(n_images), dim3(T), 0, s, images, frame_bytes, out)
  if (mode == 0 && threads == 512 && nt) L(0, 512, true);
  else if (mode == 0 && threads == 512) L(0, 512, false);
  else if (mode == 1 && threads == 512 && nt) L(1, 512, true);
  else if (mode == 1 && threads == 512) L(1, 512, false);
  else if (mode == 1 && threads == 256 && nt) L(1, 256, true);
  else if (mode == 1 && threads == 1024 && nt) L(1, 1024, true);
  else if (mode == 0 && threads == 256 && nt) L(0, 256, true);
  else return -1;
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
