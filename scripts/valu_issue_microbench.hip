// valu_issue_microbench.hip — what does one wave64 instruction cost on gfx950, alone and beside other waves?
//
// Settles the question VERDICT r1 raised about DESIGN.md's issue model ("a wave64 instruction occupies the SIMD for
// 4 cycles" vs the guide's SIMD-32 / 2 cycles): for each instruction class the step kernel is made of, measures
//   * the cycles between two DEPENDENT instructions of one wave (latency a lone dependent chain pays), and
//   * the cycles between two INDEPENDENT instructions of one wave (issue cost),
// with 1, 2, 3, 4 and 8 waves resident on every SIMD, and from them the instructions per cycle a SIMD sustains.
//
// Placement is forced, not assumed: one workgroup of 256*W threads per CU (a 96 KB LDS request keeps a second one
// out), whose 4*W waves the dispatcher deals round-robin over the CU's four SIMDs -> exactly W waves per SIMD.
// Timing: s_memtime (shader clock) around the instruction block inside each wave; s_memrealtime (100 MHz) gives the
// clock the run actually had.
//
//   hipcc --offload-arch=gfx950 -O2 -o valu_issue_microbench scripts/valu_issue_microbench.hip
//   ./valu_issue_microbench [out.json]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int kUnroll = 64;     // instructions per asm block
constexpr int kLoops = 200;     // blocks per measurement

#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)
#define REP8(x) REP4(x) REP4(x)

enum Test { FMA_DEP = 0, FMA_IND8, DPP_DEP, DPP_IND8, SWIZZLE_DEP, SWIZZLE_IND8, LDS_DEP, LDS_IND8, RCP_DEP, READLANE_DEP,
            MIX_STEP, PK_DEP, PK_IND8, SALU_DEP, SNOP, FMA_SALU_MIX, FMA_LDS_MIX, N_TESTS };
static const char* kNames[N_TESTS] = {
    "v_fma_f32 dependent chain", "v_fma_f32 8 independent chains", "v_add_f32 dpp quad_perm dependent chain",
    "v_add_f32 dpp 8 independent chains", "ds_swizzle_b32 dependent chain", "ds_swizzle_b32 8 independent",
    "ds_read_b32 dependent chain (pointer chase)", "ds_read_b32 8 independent", "v_rcp_f32 dependent chain",
    "v_readlane_b32 -> v_mov dependent chain", "ABA-like mix: 6 fma + 3 (s_nop + dpp add) + rcp + mul + 6 swizzle + 6 fma, dependent (23 counted)",
    "v_pk_fma_f32 dependent chain (2 fma per lane each)", "v_pk_fma_f32 8 independent chains", "s_add_u32 dependent chain",
    "s_nop 0", "v_fma_f32 + s_add_u32 alternating, independent (per pair)", "v_fma_f32 + ds_read_b32 alternating, independent (per pair)"};
// instructions counted per asm block for each test
static const int kPerBlock[N_TESTS] = {64, 64, 64, 64, 64, 64, 64, 64, 64, 64 * 2, 4 * 23, 64, 64, 64, 64, 64, 64};

template <int TEST>
__global__ void __launch_bounds__(1024) bench(unsigned long long* cycles, unsigned long long* real, float* sink) {
  extern __shared__ int lds[];
  const int tid = threadIdx.x;
  for (int i = tid; i < 4096; i += blockDim.x) lds[i] = ((i * 17 + 5) & 1023) * 4;   // byte offsets: a 1024-entry cycle
  __syncthreads();
  float a0 = tid * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  const float b = 0.999f, c = 1e-3f;
  int p0 = (tid & 1023) * 4, p1 = p0 ^ 64, p2 = p0 ^ 128, p3 = p0 ^ 192, p4 = p0 ^ 256, p5 = p0 ^ 320, p6 = p0 ^ 384, p7 = p0 ^ 448;
  int si = 0;
  double d0 = tid * 1e-3, d1 = d0 + 1, d2 = d0 + 2, d3 = d0 + 3, d4 = d0 + 4, d5 = d0 + 5, d6 = d0 + 6, d7 = d0 + 7;   // 64-bit pairs = two floats each
  const double db = 0.999, dc = 1e-3;
  __syncthreads();
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < kLoops; ++it) {
    if constexpr (TEST == FMA_DEP) {
      asm volatile(REP64("v_fma_f32 %0, %0, %1, %2\n") : "+v"(a0) : "v"(b), "v"(c));
    } else if constexpr (TEST == FMA_IND8) {
      asm volatile(REP8("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                        "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
    } else if constexpr (TEST == DPP_DEP) {
      // a DPP read of a VGPR the previous VALU instruction wrote needs 2 wait states (the compiler emits s_nop 1 or
      // fills them): the chain carries them, as compiled code would
      asm volatile(REP64("s_nop 1\n v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n") : "+v"(a0));
    } else if constexpr (TEST == DPP_IND8) {
      asm volatile(REP8("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                        "v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                        "v_add_f32_dpp %2, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                        "v_add_f32_dpp %3, %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                        "v_add_f32_dpp %4, %4, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                        "v_add_f32_dpp %5, %5, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                        "v_add_f32_dpp %6, %6, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                        "v_add_f32_dpp %7, %7, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    } else if constexpr (TEST == SWIZZLE_DEP) {
      asm volatile(REP64("ds_swizzle_b32 %0, %0 offset:0x0038\n s_waitcnt lgkmcnt(0)\n") : "+v"(a0));
    } else if constexpr (TEST == SWIZZLE_IND8) {
      asm volatile(REP8("ds_swizzle_b32 %0, %0 offset:0x0038\n ds_swizzle_b32 %1, %1 offset:0x0038\n ds_swizzle_b32 %2, %2 offset:0x0038\n"
                        "ds_swizzle_b32 %3, %3 offset:0x0038\n ds_swizzle_b32 %4, %4 offset:0x0038\n ds_swizzle_b32 %5, %5 offset:0x0038\n"
                        "ds_swizzle_b32 %6, %6 offset:0x0038\n ds_swizzle_b32 %7, %7 offset:0x0038\n s_waitcnt lgkmcnt(0)\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    } else if constexpr (TEST == LDS_DEP) {
      asm volatile(REP64("ds_read_b32 %0, %0\n s_waitcnt lgkmcnt(0)\n") : "+v"(p0));
    } else if constexpr (TEST == LDS_IND8) {
      asm volatile(REP8("ds_read_b32 %0, %0\n ds_read_b32 %1, %1\n ds_read_b32 %2, %2\n ds_read_b32 %3, %3\n"
                        "ds_read_b32 %4, %4\n ds_read_b32 %5, %5\n ds_read_b32 %6, %6\n ds_read_b32 %7, %7\n s_waitcnt lgkmcnt(0)\n")
                   : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7));
    } else if constexpr (TEST == RCP_DEP) {
      asm volatile(REP64("v_rcp_f32 %0, %0\n") : "+v"(a0));
    } else if constexpr (TEST == READLANE_DEP) {
      asm volatile(REP64("v_readlane_b32 %1, %0, 63\n v_mov_b32 %0, %1\n") : "+v"(a0), "+s"(si));
    } else if constexpr (TEST == PK_DEP) {
      asm volatile(REP64("v_pk_fma_f32 %0, %0, %1, %2\n") : "+v"(d0) : "v"(db), "v"(dc));
    } else if constexpr (TEST == PK_IND8) {
      asm volatile(REP8("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                        "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n")
                   : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(db), "v"(dc));
    } else if constexpr (TEST == SALU_DEP) {
      asm volatile(REP64("s_add_u32 %0, %0, 3\n") : "+s"(si) :: "scc");
    } else if constexpr (TEST == SNOP) {
      asm volatile(REP64("s_nop 0\n"));
    } else if constexpr (TEST == FMA_SALU_MIX) {
      asm volatile(REP64("v_fma_f32 %0, %0, %2, %3\n s_add_u32 %1, %1, 3\n") : "+v"(a0), "+s"(si) : "v"(b), "v"(c) : "scc");
    } else if constexpr (TEST == FMA_LDS_MIX) {
      asm volatile(REP8("v_fma_f32 %0, %0, %16, %17\n ds_read_b32 %8, %8\n v_fma_f32 %1, %1, %16, %17\n ds_read_b32 %9, %9\n"
                        "v_fma_f32 %2, %2, %16, %17\n ds_read_b32 %10, %10\n v_fma_f32 %3, %3, %16, %17\n ds_read_b32 %11, %11\n"
                        "v_fma_f32 %4, %4, %16, %17\n ds_read_b32 %12, %12\n v_fma_f32 %5, %5, %16, %17\n ds_read_b32 %13, %13\n"
                        "v_fma_f32 %6, %6, %16, %17\n ds_read_b32 %14, %14\n v_fma_f32 %7, %7, %16, %17\n ds_read_b32 %15, %15\n s_waitcnt lgkmcnt(0)\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7),
                     "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(b), "v"(c));
    } else if constexpr (TEST == MIX_STEP) {
      // one articulated-body elimination step as the kernel issues it: U = IA.s (6 dependent fma), group sum (3 dpp
      // adds), 1/D, k = U/D, six group broadcasts, six downdates — every instruction depends on the one before
      asm volatile(REP4(
          "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %0, %0, %8, %9\n"
          "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %0, %0, %8, %9\n"
          "s_nop 1\n v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
          "s_nop 1\n v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"
          "s_nop 1\n v_add_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n"
          "v_rcp_f32 %1, %0\n v_mul_f32 %1, %1, %0\n"
          "ds_swizzle_b32 %2, %1 offset:0x0018\n ds_swizzle_b32 %3, %1 offset:0x0038\n ds_swizzle_b32 %4, %1 offset:0x0058\n"
          "ds_swizzle_b32 %5, %1 offset:0x0078\n ds_swizzle_b32 %6, %1 offset:0x0098\n ds_swizzle_b32 %7, %1 offset:0x00b8\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_fma_f32 %0, %2, %1, %0\n v_fma_f32 %0, %3, %1, %0\n v_fma_f32 %0, %4, %1, %0\n v_fma_f32 %0, %5, %1, %0\n"
          "v_fma_f32 %0, %6, %1, %0\n v_fma_f32 %0, %7, %1, %0\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0) vmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  const int wave = (blockIdx.x * blockDim.x + tid) >> 6;
  if ((tid & 63) == 0) { cycles[wave] = t1 - t0; real[wave] = r1 - r0; }
  sink[blockIdx.x * blockDim.x + tid] = (float)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7) + a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(p0 + p1 + p2 + p3 + p4 + p5 + p6 + p7) + (float)si;
}

template <int TEST>
void run(int W, int n_cu, unsigned long long* d_cyc, unsigned long long* d_real, float* d_sink, double* mean_cycles, double* ghz) {
  const int threads = 256 * W;
  const size_t lds = 96 * 1024;                 // one workgroup per CU
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(bench<TEST>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  for (int rep = 0; rep < 2; ++rep) {            // first pass warms clocks and the instruction cache
    hipLaunchKernelGGL(bench<TEST>, dim3(n_cu), dim3(threads), lds, 0, d_cyc, d_real, d_sink);
    CHECK(hipDeviceSynchronize());
  }
  const int n_wave = n_cu * 4 * W;
  std::vector<unsigned long long> c(n_wave), r(n_wave);
  CHECK(hipMemcpy(c.data(), d_cyc, sizeof(unsigned long long) * n_wave, hipMemcpyDeviceToHost));
  CHECK(hipMemcpy(r.data(), d_real, sizeof(unsigned long long) * n_wave, hipMemcpyDeviceToHost));
  double sc = 0, sr = 0;
  for (int i = 0; i < n_wave; ++i) { sc += (double)c[i]; sr += (double)r[i]; }
  *mean_cycles = sc / n_wave;
  *ghz = sc / sr * 0.1;                         // s_memrealtime ticks at 100 MHz
}

int main(int argc, char** argv) {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int n_cu = prop.multiProcessorCount;
  unsigned long long *d_cyc, *d_real;
  float* d_sink;
  CHECK(hipMalloc(&d_cyc, sizeof(unsigned long long) * n_cu * 32));
  CHECK(hipMalloc(&d_real, sizeof(unsigned long long) * n_cu * 32));
  CHECK(hipMalloc(&d_sink, sizeof(float) * n_cu * 2048));
  const int Ws[] = {1, 2, 3, 4};
  std::string json = "{\n \"device\": \"" + std::string(prop.gcnArchName) + "\", \"compute_units\": " + std::to_string(n_cu) +
                     ",\n \"method\": \"one workgroup of 256*W threads per CU (96 KB LDS) = W waves on every SIMD; s_memtime around " +
                     std::to_string(kLoops) + " blocks of straight-line asm per wave; cycles_per_inst = per-wave cycles between two of its "
                     "own instructions, simd_ipc = W / cycles_per_inst\",\n \"tests\": [\n";
  printf("%-78s %5s %14s %10s %8s\n", "test", "W", "cyc/inst/wave", "SIMD IPC", "GHz");
  for (int t = 0; t < N_TESTS; ++t) {
    json += std::string("  {\"name\": \"") + kNames[t] + "\", \"by_waves_per_simd\": {";
    for (size_t wi = 0; wi < sizeof(Ws) / sizeof(Ws[0]); ++wi) {
      const int W = Ws[wi];
      double cyc = 0, ghz = 0;
      switch (t) {
#define CASE(T) case T: run<T>(W, n_cu, d_cyc, d_real, d_sink, &cyc, &ghz); break;
        CASE(FMA_DEP) CASE(FMA_IND8) CASE(DPP_DEP) CASE(DPP_IND8) CASE(SWIZZLE_DEP) CASE(SWIZZLE_IND8) CASE(LDS_DEP)
        CASE(LDS_IND8) CASE(RCP_DEP) CASE(READLANE_DEP) CASE(MIX_STEP) CASE(PK_DEP) CASE(PK_IND8) CASE(SALU_DEP) CASE(SNOP)
        CASE(FMA_SALU_MIX) CASE(FMA_LDS_MIX)
#undef CASE
      }
      const double per = cyc / ((double)kLoops * kPerBlock[t]);
      printf("%-78s %5d %14.2f %10.3f %8.2f\n", kNames[t], W, per, W / per, ghz);
      char buf[160];
      snprintf(buf, sizeof(buf), "%s\"%d\": {\"cycles_per_inst\": %.3f, \"simd_ipc\": %.4f, \"ghz\": %.3f}", wi ? ", " : "", W, per, W / per, ghz);
      json += buf;
    }
    json += std::string("}}") + (t + 1 < N_TESTS ? ",\n" : "\n");
  }
  json += " ]\n}\n";
  if (argc > 1) { FILE* f = fopen(argv[1], "w"); if (f) { fputs(json.c_str(), f); fclose(f); } }
  return 0;
}
