"""Kinematic-replay inputs: the Spotlight walking clip resampled to the simulation timestep.

Mirror of the reference's ``MotionSnippet`` (``src/flygym_demo/spotlight_data/preprocessing.py:11-142``)
and of the benchmark's ``ReplayTargetData`` (``src/flygym_demo/benchmark/time_gpu_simulation.py:67-86``).
The recorded joint angles (660 frames x 6 legs x 7 DoF at 330 Hz) come from the asset pack.
"""

from __future__ import annotations

import numpy as np

from .anatomy import JointDOF

__all__ = ["MotionSnippet", "ReplayTargetData", "savgol_taps"]


def savgol_taps(window: int, polyorder: int) -> np.ndarray:
    """Constants of ``scipy.signal.savgol_filter(mode="interp")`` as one float64 vector (the layout
    ``nmf_replay_resample`` takes): ``window`` interior taps, then ``window // 2`` rows of ``window`` taps for the first
    frames (value at frame ``i`` of the polynomial fitted to the first window), then as many rows for the last frames.
    Plain least squares: the tap for sample ``k`` when evaluating at offset ``x0`` is ``A (A^T A)^-1 x0^p``."""
    half = window // 2
    x = np.arange(-half, half + 1, dtype=np.float64)
    A = np.vander(x, polyorder + 1, increasing=True)
    pinv = A @ np.linalg.inv(A.T @ A)                       # (window, polyorder + 1)

    def at(x0):
        return pinv @ (float(x0) ** np.arange(polyorder + 1))

    # interior taps exactly as scipy.signal.savgol_coeffs obtains them (a minimum-norm least-squares solve of the moment
    # conditions, LAPACK gelsd): the same float64 values to the last bit, hence the same float32 smoothed frames
    order = np.arange(polyorder + 1).reshape(-1, 1)
    unit = np.zeros(polyorder + 1); unit[0] = 1.0
    mid = np.linalg.lstsq(x[::-1] ** order, unit, rcond=None)[0]
    rows = [mid] + [at(i - half) for i in range(half)] + [at(p + 1) for p in range(half)]
    return np.concatenate(rows)


class MotionSnippet:
    def __init__(self, data_path=None, *, angles_global2anatomical: bool = True) -> None:
        if data_path is None:
            from .compiler.model import load_asset_pack

            pack = load_asset_pack()
            self.joint_angles = pack["clip_joint_angles"].astype(np.float32).copy()
            self.legs = [str(x) for x in pack["clip_legs"]]
            self.dofs_per_leg = [tuple(str(y) for y in row) for row in pack["clip_dofs_per_leg"]]
            self.data_fps = float(pack["clip_fps"])
            self.rawpred_egoxyz = pack["clip_rawpred_egoxyz"].copy()
            self.fwdkin_egoxyz = pack["clip_fwdkin_egoxyz"].copy()
            self.keypoints = [tuple(str(y) for y in row) for row in pack["clip_keypoints"]]
            self.experiment_trial = str(pack["clip_experiment_trial"])
            self.framerange_in_raw_recording = [int(x) for x in pack["clip_framerange"]]
        else:
            data = np.load(data_path, allow_pickle=True)
            self.joint_angles = data["joint_angles"].copy()
            self.legs = data["legs"].tolist()
            self.dofs_per_leg = [tuple(x) for x in data["dofs_per_leg"].tolist()]
            self.data_fps = float(data["data_fps"].item())
            self.rawpred_egoxyz, self.fwdkin_egoxyz = data["rawpred_egoxyz"], data["fwdkin_egoxyz"]
            self.keypoints = [tuple(x) for x in data["keypoints"].tolist()]
            self.experiment_trial = data["experiment_trial"].item()
            self.framerange_in_raw_recording = data["framerange_in_raw_recording"].tolist()
        if angles_global2anatomical:
            # right-leg roll / yaw change sign: global (IK) convention -> anatomical (:59-78)
            right = [i for i, leg in enumerate(self.legs) if leg[0] == "r"]
            flip = [i for i, (_, _, axis) in enumerate(self.dofs_per_leg) if axis in ("roll", "yaw")]
            self.joint_angles[np.ix_(np.arange(self.joint_angles.shape[0]), right, flip)] *= -1

    def get_joint_angles(self, output_timestep: float, output_dof_order: list[JointDOF], *,
                         sgfilter_window_sec: float = 0.03, sgfilter_polyorder: int = 3) -> np.ndarray:
        """Savitzky-Golay smoothing, cubic resampling onto ``arange(0, T, dt)``, reorder (:80-142)."""
        from scipy.interpolate import interp1d
        from scipy.signal import savgol_filter

        window = int(sgfilter_window_sec * self.data_fps)
        window += 1 - (window % 2)
        smooth = savgol_filter(self.joint_angles, window_length=window, polyorder=sgfilter_polyorder, axis=0)
        n = self.joint_angles.shape[0]
        t_src = np.arange(n) / self.data_fps
        t_out = np.arange(0, n / self.data_fps, output_timestep)
        interp = interp1d(t_src, smooth, kind="cubic", axis=0, bounds_error=False,
                          fill_value=(smooth[0], smooth[-1]))
        dense = interp(t_out)
        cols = [
            (self.legs.index(d.child.pos), self.dofs_per_leg.index((d.parent.link, d.child.link, d.axis.value)))
            for d in output_dof_order
        ]
        legs, dofs = np.array(cols, dtype=np.int64).T
        return dense[:, legs, dofs]

    def get_joint_angles_device(self, output_timestep: float, output_dof_order: list[JointDOF], device, *,
                                sgfilter_window_sec: float = 0.03, sgfilter_polyorder: int = 3):
        """The same table computed on the GPU (``nmf_replay_resample``: Savitzky-Golay + not-a-knot cubic spline in
        float64, one workgroup per column): float32 torch tensor ``(n_output_steps, len(output_dof_order))`` on
        ``device``.  Only the clip (660 x 42 floats) and the filter constants cross PCIe.

        The kernel keeps a column of the clip in LDS: clips of 6 to 1536 frames with a filter window no longer than the
        clip.  Anything else takes the host path (:meth:`get_joint_angles`, scipy) and is copied to ``device``."""
        import torch

        from . import _native

        window = int(sgfilter_window_sec * self.data_fps)
        window += 1 - (window % 2)
        n_frames = self.joint_angles.shape[0]
        if not 6 <= n_frames <= 1536 or window > n_frames or window < 3:
            host = self.get_joint_angles(output_timestep, output_dof_order, sgfilter_window_sec=sgfilter_window_sec,
                                         sgfilter_polyorder=sgfilter_polyorder)
            return torch.as_tensor(np.ascontiguousarray(host, dtype=np.float32), device=device)
        cols = [
            (self.legs.index(d.child.pos), self.dofs_per_leg.index((d.parent.link, d.child.link, d.axis.value)))
            for d in output_dof_order
        ]
        legs, dofs = np.array(cols, dtype=np.int64).T
        n = self.joint_angles.shape[0]
        clip = torch.as_tensor(np.ascontiguousarray(self.joint_angles[:, legs, dofs], dtype=np.float32), device=device)
        taps = torch.as_tensor(savgol_taps(window, sgfilter_polyorder), device=device)
        n_out = len(np.arange(0, n / self.data_fps, output_timestep))
        out = torch.empty((n_out, len(cols)), dtype=torch.float32, device=device)
        with torch.cuda.device(clip.device):
            stream = torch.cuda.current_stream(clip.device).cuda_stream
            _native.check(_native.lib().nmf_replay_resample(clip.data_ptr(), n, len(cols), float(self.data_fps),
                                                            float(output_timestep), taps.data_ptr(), window, n_out,
                                                            out.data_ptr(), stream))
        return out


class ReplayTargetData:
    """World ``w`` replays clip partition ``w % n_partitions`` (benchmark :73-86)."""

    def __init__(self, sim_timestep: float, output_dof_order: list[JointDOF], device=None):
        """``device``: build the table on that GPU (``MotionSnippet.get_joint_angles_device``) instead of with scipy
        on the host; ``make_target_angles_all_worlds`` then returns a device tensor."""
        self.snippet = MotionSnippet()
        self.device = device
        if device is None:
            self.dof_angles = self.snippet.get_joint_angles(sim_timestep, output_dof_order)
        else:
            self.dof_angles = self.snippet.get_joint_angles_device(sim_timestep, output_dof_order, device)
        self.n_total_steps, self.n_dofs = self.dof_angles.shape

    def make_target_angles_all_worlds(self, n_worlds: int, sim_steps: int, first_world: int = 0) -> np.ndarray:
        n_partitions = self.n_total_steps // sim_steps
        if self.device is not None:
            import torch

            part = (first_world + torch.arange(n_worlds, device=self.dof_angles.device)) % n_partitions
            idx = part[:, None] * sim_steps + torch.arange(sim_steps, device=self.dof_angles.device)[None, :]
            return self.dof_angles[idx].contiguous()
        part = (first_world + np.arange(n_worlds)) % n_partitions
        idx = part[:, None] * sim_steps + np.arange(sim_steps)[None, :]
        return np.ascontiguousarray(self.dof_angles[idx].astype(np.float32))
