"""CPU restatement (numpy float64) of the device replay resample ``nmf_replay_resample`` (flygym_amd/csrc/nmf_replay.hip).
TEST INFRASTRUCTURE: the same three steps as the kernel, written with explicit recurrences, so that the kernel can be
checked against it — and it against the reference's scipy pipeline (``src/flygym_demo/spotlight_data/preprocessing.py:
80-142``, reproduced in ``flygym_amd.replay.MotionSnippet.get_joint_angles`` and frozen in tests/golden/replay_42.npz)."""

import numpy as np


def resample(clip: np.ndarray, fps: float, out_dt: float, taps: np.ndarray, window: int) -> np.ndarray:
    """clip (n_frames, n_cols) -> float32 (n_out, n_cols)."""
    x = np.asarray(clip, dtype=np.float32).astype(np.float64)
    n, half = x.shape[0], window // 2
    y = np.zeros_like(x)
    for i in range(n):
        if i < half:
            y[i] = taps[window + i * window: window + (i + 1) * window] @ x[:window]
        elif i >= n - half:
            p = i - (n - half)
            y[i] = taps[window + half * window + p * window: window + half * window + (p + 1) * window] @ x[n - window:]
        else:
            y[i] = taps[:window] @ x[i - half: i + half + 1]
    # the reference filters a float32 array: scipy accumulates in float64 and stores float32 (savgol_filter keeps the
    # input's single precision), so the spline below goes through the float32-rounded smoothed frames
    y = y.astype(np.float32).astype(np.float64)
    h = 1.0 / fps
    s = 6.0 / (h * h)
    rhs = lambda i: s * (y[i - 1] - 2.0 * y[i] + y[i + 1])
    M = np.zeros_like(y)
    cp = np.zeros(n)
    M[1] = rhs(1) / 6.0
    M[n - 2] = rhs(n - 2) / 6.0
    for i in range(2, n - 2):
        r = rhs(i)
        if i == 2:
            r = r - M[1]
        if i == n - 3:
            r = r - M[n - 2]
        denom = 4.0 if i == 2 else 4.0 - cp[i - 1]
        cp[i] = 1.0 / denom
        M[i] = (r if i == 2 else r - M[i - 1]) / denom
    for i in range(n - 4, 1, -1):
        M[i] -= cp[i] * M[i + 1]
    M[0] = 2.0 * M[1] - M[2]
    M[n - 1] = 2.0 * M[n - 2] - M[n - 3]
    t = np.arange(0, n / fps, out_dt)
    x_last = (n - 1) / fps
    i = np.minimum((t * fps).astype(np.int64), n - 2)
    for _ in range(2):
        i = np.where((i < n - 2) & ((i + 1) / fps <= t), i + 1, i)
        i = np.where((i > 0) & (i / fps > t), i - 1, i)
    u = (t - i / fps) / h
    w = 1.0 - u
    v = w[:, None] * y[i] + u[:, None] * y[i + 1] + (h * h / 6.0) * ((w * w * w - w)[:, None] * M[i] + (u * u * u - u)[:, None] * M[i + 1])
    v = np.where((t > x_last)[:, None], y[n - 1][None, :], v)
    return v.astype(np.float32)
