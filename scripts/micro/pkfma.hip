// microbenchmark: issue rate of v_fma_f32 against v_pk_fma_f32 on gfx950 (waves of 64, 8 waves per SIMD)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int PK> __global__ void __launch_bounds__(256) k(float* out, int n, float a, float b) {
  float x[8]; f2 y[8];
  for (int i = 0; i < 8; ++i) { x[i] = threadIdx.x * 1e-3f + i; y[i] = f2{x[i], x[i] + 1.f}; }
  f2 a2 = {a, a}, b2 = {b, b};
  for (int it = 0; it < n; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (PK) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(y[i]) : "v"(a2), "v"(b2));
        else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
      }
  }
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += PK ? y[i].x + y[i].y : x[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
  float* out; hipMalloc(&out, 256 * 2048 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int n = 4000, blocks = 2048;
  for (int pk = 0; pk < 2; ++pk) for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    if (pk) k<1><<<blocks, 256>>>(out, n, 0.999f, 0.001f); else k<0><<<blocks, 256>>>(out, n, 0.999f, 0.001f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double inst = (double)blocks * 4 * n * 64;    // wave-instructions
    double fl = inst * 64 * 2 * (pk ? 2 : 1);
    printf("pk=%d  %.3f ms  %.2f Tinst(wave)/s  %.1f TFLOP/s  cycles per wave-inst per SIMD at 2.4 GHz: %.2f\n", pk, ms, inst / ms / 1e9, fl / ms / 1e9,
           ms * 1e-3 * 2.4e9 * 1024 / inst);
  }
  return 0;
}
