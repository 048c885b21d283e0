// Micro-benchmark of the articulated-body sweep in isolation (diagnostic; built and run by scripts/aba_microbench.py)
#include "nmf_capi.hip"
namespace nmf {
template <class TP>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2)))
aba_bench_kernel(const DevModel* mp, DevState st, unsigned long long* cycles, int reps, int withK) {
  __shared__ FlyLds<TP> s;
  const DevModel& m = *mp;
  const int w = blockIdx.x, lane = threadIdx.x;
  for (int j = lane; j < TP::NV; j += kWave) { s.arm[j] = m.dof_armature[j]; s.damp[j] = m.dof_damping[j]; }
  for (int i = lane; i < TP::NQ; i += kWave) s.qpos[i] = st.qpos[(size_t)w * TP::NQ + i];
  for (int i = lane; i < TP::NV; i += kWave) { s.qvel[i] = st.qvel[(size_t)w * TP::NV + i]; s.qacc[i] = 0.f; }
  for (int i = lane; i < m.nu; i += kWave) s.ctrl[i] = st.ctrl[(size_t)w * m.nu + i];
  __syncthreads();
  const Frame fr = make_frame(v3(m.plane[0], m.plane[1], m.plane[2]));
  stage_kinematics(s, m, lane);
  stage_inertia(s, m, lane);
  stage_collision(s, m, lane);
  if (lane < s.ncon) { s.c_mu[lane] = 1.f; s.c_D[lane] = 1e-3f; s.c_info[lane] |= (withK ? 0xf : 0) << 20; }
  for (int j = lane; j < TP::NV; j += kWave) s.vA[j] = 0.01f * (float)(j % 7) - 0.02f;
  __syncthreads();
  unsigned long long t0 = clock64();
  for (int r = 0; r < reps; ++r) {
    aba_solve<TP, false>(s, V_A, V_B, withK != 0, 0.f, m, lane);
    for (int j = lane; j < TP::NV; j += kWave) s.vA[j] += 1e-3f * s.vB[j];
    __syncthreads();
  }
  unsigned long long t1 = clock64();
  if (lane == 0) cycles[w] = (t1 - t0) / reps;
  for (int j = lane; j < TP::NV; j += kWave) st.qacc[(size_t)w * TP::NV + j] = s.vB[j];
}
}  // namespace nmf
extern "C" int nmf_aba_bench(nmf_batch* b, unsigned long long* cycles_dev, int reps, int withK) {
  hipLaunchKernelGGL((nmf::aba_bench_kernel<nmf::FlyTopo>), dim3((unsigned)b->n_worlds), dim3(64), 0, 0, b->dm_dev, b->st, cycles_dev, reps, withK);
  return hipDeviceSynchronize() == hipSuccess ? 0 : -1;
}
