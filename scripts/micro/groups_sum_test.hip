// checks groups_sum (nmf_device.h): lane r of every 8-lane group must end with the sum over the eight groups of lane r
#include <hip/hip_runtime.h>
#include <cstdio>
#define NMF_DPP(v, ctrl) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (v)), (ctrl), 0xf, 0xf, true))
__global__ void k(float* out) {
  const int l = threadIdx.x;
  float v = (float)(1 << (l >> 3));
  v += NMF_DPP(v, 0x128);
  out[l] = v;
  { float a = v, b = v; asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    out[64 + l] = a; out[128 + l] = b;
    v = a + b; }
  out[192 + l] = v;
  { float a = v, b = v; asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    out[256 + l] = a; out[320 + l] = b;
    v = a + b; }
  out[384 + l] = v;
}
int main() {
  float* d; hipMalloc(&d, 448 * 4); k<<<1, 64>>>(d); float h[448]; hipMemcpy(h, d, 448 * 4, hipMemcpyDeviceToHost);
  for (int b = 0; b < 7; ++b) { for (int l = 0; l < 64; l += 8) printf("%g ", h[64 * b + l]); printf("\n"); }
  return 0;
}
