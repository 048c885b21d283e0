"""Open-loop controllers that produce position-actuator targets for the batched engine.

``TripodCPG`` is the "position-actuated CPG tripod gait" of BASELINE config 2.  The reference snapshot has
no CPG (flygym 2.0.1 dropped flygym 1.x's controllers), so this one is build-defined:

* six phase oscillators, one per leg, advancing at ``frequency`` (default 12 Hz); tripod phase biases
  ``{lf, rm, lh} = 0`` and ``{rf, lm, rh} = pi``;
* each leg's seven actuated joint angles are a periodic function of its phase: one step cycle cut from the
  Spotlight walking clip (the mean stride between the clip's swing onsets of that leg), resampled on a uniform
  phase grid and blended to be periodic;
* world ``w`` of ``n_worlds`` starts with the global phase offset ``2 pi w / n_worlds`` — deterministic, seed free.

The controller emits a ``(n_worlds, steps, 42)`` float32 target table on the GPU, i.e. exactly the input of
``HIPSimulation.step_replay`` / ``nmf_step_replay``: the CPG runs inside the stepping kernel's control-load stage.

Leg adhesion driven by the gait (BASELINE config 5): ``stance_bins`` marks, per leg, the part of the step cycle in
which the claw is near its lowest point (forward kinematics of the cycle in the thorax frame); with
``adhesion=(stance, on, off)`` the table gets six more columns holding the adhesion control ``on`` in stance and
``off`` in swing.  The reference clamps adhesion controls to [1, 100] (``compose/fly.py:434-440``), so "off" is 1.
"""

from __future__ import annotations

import numpy as np

from .anatomy import LEGS, JointDOF
from .replay import MotionSnippet

__all__ = ["TripodCPG"]

TRIPOD_PHASE_BIAS = {"lf": 0.0, "rm": 0.0, "lh": 0.0, "rf": np.pi, "lm": np.pi, "rh": np.pi}


class TripodCPG:
    def __init__(self, actuated_dofs: list[JointDOF], timestep: float, *, frequency: float = 12.0, n_phase_bins: int = 256):
        self.actuated_dofs = list(actuated_dofs)
        self.timestep = float(timestep)
        self.frequency = float(frequency)
        self.n_bins = int(n_phase_bins)
        # (T, n) at the sim timestep.  Dofs the walking clip does not have (ALL_POSSIBLE's extra axes of the leg joints) are
        # held at zero, their neutral angle.
        snippet = MotionSnippet()
        in_clip = [(d.parent.link, d.child.link, d.axis.value) in snippet.dofs_per_leg for d in self.actuated_dofs]
        known = [d for d, k in zip(self.actuated_dofs, in_clip) if k]
        part = snippet.get_joint_angles(timestep, known)
        clip = np.zeros((part.shape[0], len(self.actuated_dofs)), dtype=part.dtype)
        clip[:, np.nonzero(in_clip)[0]] = part
        self.leg_of_dof = np.array([LEGS.index(d.child.pos) for d in self.actuated_dofs])
        self.cycle = np.zeros((self.n_bins, len(self.actuated_dofs)), dtype=np.float32)
        for leg in range(6):
            cols = np.where(self.leg_of_dof == leg)[0]
            knee = [i for i, c in enumerate(cols) if (self.actuated_dofs[c].parent.link, self.actuated_dofs[c].child.link,
                                                      self.actuated_dofs[c].axis.value) == ("trochanterfemur", "tibia", "pitch")]
            self.cycle[:, cols] = self._step_cycle(clip[:, cols], knee[0] if knee else min(5, len(cols) - 1))

    def _step_cycle(self, angles: np.ndarray, key_col: int) -> np.ndarray:
        """One periodic stride of a leg: stance/swing onsets from the trochanterfemur-tibia pitch angle's
        upward mean crossings, strides resampled to ``n_bins`` phase bins and averaged."""
        key = angles[:, key_col]
        centred = key - key.mean()
        onsets = np.where((centred[:-1] < 0) & (centred[1:] >= 0))[0]
        onsets = onsets[np.diff(onsets, prepend=-10 ** 9) > int(0.02 / self.timestep)]   # debounce 20 ms
        grid = np.linspace(0.0, 1.0, self.n_bins, endpoint=False)
        strides = []
        for a, b in zip(onsets[:-1], onsets[1:]):
            if b - a < int(0.03 / self.timestep):
                continue
            src = np.linspace(0.0, 1.0, b - a, endpoint=False)
            strides.append(np.stack([np.interp(grid, src, angles[a:b, k]) for k in range(angles.shape[1])], axis=1))
        if not strides:
            raise ValueError("no stride found in the clip")
        cyc = np.mean(strides, axis=0)
        # make the cycle periodic: remove the end-to-start jump linearly over the cycle
        jump = cyc[0] - (2 * cyc[-1] - cyc[-2])
        cyc = cyc + np.outer(grid, jump)
        return cyc.astype(np.float32)

    def phases(self, n_worlds: int, steps: int, start_step: int = 0, first_world: int = 0,
               total_worlds: int | None = None) -> np.ndarray:
        """(n_worlds, steps, 6) oscillator phases in [0, 2 pi); ``first_world`` / ``total_worlds`` place a
        shard of worlds inside a larger (multi-GPU) population."""
        t = (start_step + np.arange(steps)) * self.timestep
        world = 2 * np.pi * (first_world + np.arange(n_worlds)) / (total_worlds or n_worlds)
        bias = np.array([TRIPOD_PHASE_BIAS[leg] for leg in LEGS])
        ph = 2 * np.pi * self.frequency * t[None, :, None] + world[:, None, None] + bias[None, None, :]
        return np.mod(ph, 2 * np.pi)

    def stance_bins(self, model, fly, threshold: float = 0.3) -> np.ndarray:
        """``(n_bins, 6)`` bool: leg ``l`` is in stance in phase bin ``i`` when the origin of its last tarsal segment,
        computed by forward kinematics of the step cycle with the thorax at the identity pose, lies within
        ``threshold`` of its height range above its lowest point.  ``model`` is the compiled model of a world
        holding ``fly`` (``HIPSimulation.model``)."""
        from .compiler.rigid import forward_kinematics

        pos_ids = [i for i, a in enumerate(fly.actuators) if a["kind"] == "position"]
        if len(pos_ids) != len(self.actuated_dofs):
            raise ValueError("the fly's position actuators do not match the controller's actuated dofs")
        qadr = np.asarray(model["act_trn"])[pos_ids] + 1
        segs = [s.name for s in fly.get_bodysegs_order()]
        claw_body = [int(model["seg_body"][segs.index(f"{leg}_tarsus5")]) for leg in LEGS]
        q = np.array(model["key_qpos"], dtype=np.float64)
        q[:7] = (0, 0, 0, 1, 0, 0, 0)
        z = np.zeros((self.n_bins, 6))
        for i in range(self.n_bins):
            q[qadr] = self.cycle[i]
            xpos, _, _ = forward_kinematics(model, q)
            z[i] = xpos[claw_body, 2]
        lo, hi = z.min(axis=0), z.max(axis=0)
        return z <= lo + threshold * (hi - lo)

    def targets(self, n_worlds: int, steps: int, start_step: int = 0, device=None, first_world: int = 0,
                total_worlds: int | None = None, adhesion=None):
        """Target table ``(n_worlds, steps, n_act)`` float32: a torch tensor built on ``device`` if given (no large
        host arrays), else numpy.  ``adhesion=(stance_bins, on, off)`` appends six adhesion-control columns
        (legs in ``LEGS`` order): ``on`` while the leg's phase bin is a stance bin, else ``off``."""
        if adhesion is not None:
            stance, on, off = adhesion
            pos = self.targets(n_worlds, steps, start_step, device, first_world, total_worlds)
            legbias = np.array([TRIPOD_PHASE_BIAS[leg] for leg in LEGS])
            if device is None:
                ph = self.phases(n_worlds, steps, start_step, first_world, total_worlds)
                idx = np.floor(ph / (2 * np.pi) * self.n_bins).astype(np.int64) % self.n_bins
                adh = np.where(np.asarray(stance)[idx, np.arange(6)[None, None, :]], on, off).astype(np.float32)
                return np.ascontiguousarray(np.concatenate([pos, adh], axis=2))
            import torch

            st = torch.as_tensor(np.asarray(stance), device=device)
            t = (start_step + torch.arange(steps, device=device, dtype=torch.float64)) * self.timestep * self.frequency
            w = (first_world + torch.arange(n_worlds, device=device, dtype=torch.float64)) / float(total_worlds or n_worlds)
            b = torch.as_tensor(legbias / (2 * np.pi), device=device, dtype=torch.float64)
            x = torch.remainder(t[None, :, None] + w[:, None, None] + b[None, None, :], 1.0) * self.n_bins
            idx = torch.floor(x).to(torch.int64) % self.n_bins
            adh = torch.where(st[idx, torch.arange(6, device=device)[None, None, :]],
                              torch.tensor(float(on), device=device), torch.tensor(float(off), device=device))
            return torch.cat([pos, adh.to(torch.float32)], dim=2).contiguous()
        n_act = len(self.actuated_dofs)
        bias = np.array([TRIPOD_PHASE_BIAS[leg] for leg in LEGS])[self.leg_of_dof]                 # (n_act,)
        if device is None:
            ph = self.phases(n_worlds, steps, start_step, first_world, total_worlds)[..., self.leg_of_dof]
            x = ph / (2 * np.pi) * self.n_bins
            i0 = np.floor(x).astype(np.int64) % self.n_bins
            frac = (x - np.floor(x)).astype(np.float32)
            cols = np.arange(n_act)[None, None, :]
            table = (1 - frac) * self.cycle[i0, cols] + frac * self.cycle[(i0 + 1) % self.n_bins, cols]
            return np.ascontiguousarray(table.astype(np.float32))
        import torch

        # phase in cycles, float64 on the device for the same rounding as the numpy path, world chunks bound memory
        cyc = torch.as_tensor(self.cycle, device=device)                                            # (bins, n_act)
        t = (start_step + torch.arange(steps, device=device, dtype=torch.float64)) * self.timestep * self.frequency
        b = torch.as_tensor(bias / (2 * np.pi), device=device, dtype=torch.float64)
        out = torch.empty((n_worlds, steps, n_act), dtype=torch.float32, device=device)
        cols = torch.arange(n_act, device=device)[None, None, :]
        chunk = max(1, (1 << 24) // max(1, steps * n_act))
        for w0 in range(0, n_worlds, chunk):
            w = torch.arange(w0, min(n_worlds, w0 + chunk), device=device, dtype=torch.float64)
            world = (first_world + w) / float(total_worlds or n_worlds)
            x = torch.remainder(t[None, :, None] + world[:, None, None] + b[None, None, :], 1.0) * self.n_bins
            i0 = torch.floor(x).to(torch.int64) % self.n_bins
            frac = (x - torch.floor(x)).to(torch.float32)
            out[w0:w0 + len(w)] = (1 - frac) * cyc[i0, cols] + frac * cyc[(i0 + 1) % self.n_bins, cols]
        return out
