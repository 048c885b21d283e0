// nmf_replay.hip — the kinematic-replay data path on the device (gfx950).  SURVEY §8 (f) rank 4.
//
// Reference: MotionSnippet.get_joint_angles (src/flygym_demo/spotlight_data/preprocessing.py:80-142) smooths the
// recorded joint angles with a Savitzky-Golay filter (scipy savgol_filter, mode "interp": fixed taps in the interior,
// a polynomial fitted to the first / last window evaluated at the edge frames) and resamples them onto the simulation
// time grid with a cubic interpolant (scipy interp1d kind="cubic" = the not-a-knot cubic spline through all frames),
// holding the last frame beyond the last knot.  Here the same three linear steps run in one kernel, in float64 like the
// reference, one workgroup per joint-angle column; only the final cast is float32 (the control table's type).
//
//   smooth = SG(clip)                      interior taps / edge rows are constants the host passes in (sg_taps)
//   M      = second derivatives of the not-a-knot spline (uniform knots: M1 and M(n-2) follow directly from the
//            not-a-knot conditions, the rest is a tridiagonal solve — Thomas algorithm, one lane)
//   out[k] = spline(k * out_dt)            piecewise cubic from (y, M) of the interval that holds k * out_dt
#include "nmf_device.h"

namespace nmf {

constexpr int kReplayThreads = 256;
constexpr int kReplayMaxFrames = 1536;      // 3 x 12 KB of float64 in LDS

__global__ void __launch_bounds__(kReplayThreads)
nmf_replay_resample_kernel(const float* __restrict__ clip, int n_frames, int n_cols, double fps, double out_dt,
                           const double* __restrict__ taps, int window, int n_out, float* __restrict__ out) {
  __shared__ double y[kReplayMaxFrames], M[kReplayMaxFrames], cp[kReplayMaxFrames];
  const int col = blockIdx.x, half = window / 2;
  const double* tap_mid = taps;
  const double* tap_first = taps + window;                 // [half][window]: frame p < half from the first window
  const double* tap_last = taps + window + half * window;  // [half][window]: frame n - half + p from the last window
  // ---- Savitzky-Golay
  for (int i = threadIdx.x; i < n_frames; i += kReplayThreads) {
    double acc = 0.0;
    if (i < half) {
      for (int k = 0; k < window; ++k) acc += tap_first[i * window + k] * (double)clip[(size_t)k * n_cols + col];
    } else if (i >= n_frames - half) {
      const int p = i - (n_frames - half);
      for (int k = 0; k < window; ++k) acc += tap_last[p * window + k] * (double)clip[(size_t)(n_frames - window + k) * n_cols + col];
    } else {
      for (int k = 0; k < window; ++k) acc += tap_mid[k] * (double)clip[(size_t)(i - half + k) * n_cols + col];
    }
    // the reference filters a float32 array: scipy accumulates in float64 and stores float32 (savgol_filter keeps its
    // input's single precision), so the spline goes through the float32-rounded smoothed frames
    y[i] = (double)(float)acc;
  }
  __syncthreads();
  // ---- not-a-knot cubic spline, uniform knots h = 1 / fps: second derivatives
  const double h = 1.0 / fps;
  if (threadIdx.x == 0) {
    const int n = n_frames;
    const double s = 6.0 / (h * h);
    auto rhs = [&](int i) { return s * (y[i - 1] - 2.0 * y[i] + y[i + 1]); };
    // not-a-knot: M0 - 2 M1 + M2 = 0 folded into row 1 gives 6 M1 = rhs(1); likewise at the other end
    M[1] = rhs(1) / 6.0;
    M[n - 2] = rhs(n - 2) / 6.0;
    // rows 2 .. n-3:  M[i-1] + 4 M[i] + M[i+1] = rhs(i)  with M[1], M[n-2] known — Thomas algorithm (d' in M, c' in cp)
    for (int i = 2; i <= n - 3; ++i) {
      double r = rhs(i);
      if (i == 2) r -= M[1];
      if (i == n - 3) r -= M[n - 2];
      const double denom = i == 2 ? 4.0 : 4.0 - cp[i - 1];
      cp[i] = 1.0 / denom;
      M[i] = (i == 2 ? r : r - M[i - 1]) / denom;
    }
    for (int i = n - 4; i >= 2; --i) M[i] -= cp[i] * M[i + 1];
    M[0] = 2.0 * M[1] - M[2];
    M[n - 1] = 2.0 * M[n - 2] - M[n - 3];
  }
  __syncthreads();
  // ---- evaluation on the output grid t = k * out_dt
  const double x_last = (double)(n_frames - 1) / fps;
  const double h2_6 = h * h / 6.0;
  for (int k = threadIdx.x; k < n_out; k += kReplayThreads) {
    const double t = (double)k * out_dt;
    double v;
    if (t > x_last) v = y[n_frames - 1];                       // fill value beyond the last knot
    else {
      int i = (int)(t * fps);
      if (i > n_frames - 2) i = n_frames - 2;
      while (i < n_frames - 2 && (double)(i + 1) / fps <= t) ++i;   // knots are arange(n) / fps, compared exactly
      while (i > 0 && (double)i / fps > t) --i;
      const double u = (t - (double)i / fps) / h, w = 1.0 - u;
      v = w * y[i] + u * y[i + 1] + h2_6 * ((w * w * w - w) * M[i] + (u * u * u - u) * M[i + 1]);
    }
    out[(size_t)k * n_cols + col] = (float)v;
  }
}

}  // namespace nmf
