"""CPU check of the eye renderer's capsule culling: exact (some ray of the chunk hits the capsule's bounding shapes) against the
conservative cone tests (CPU only; poses from the oracle)."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "oracle"))
import numpy as np
import oracle as orc, sensors_oracle as so
from flygym_amd import make_model
from flygym_amd.vision import body_capsules, _euler_xyz_extrinsic_quat
from flygym_amd.sensors import EYE_CAMERAS
fly, world, _ = make_model()
m = world.compile_model()
o = orc.Oracle(m.to_blob(), "f64"); o.ctrl[42:] = 1.0; o.step(300)
names = [s.name for s in fly.get_bodysegs_order()]
xpos = o.arr("seg_xpos").reshape(-1, 3); xquat = o.arr("seg_xquat").reshape(-1, 4)
seg, geom = body_capsules(fly)
h, w, fov = 512, 450, 157.0
half_fov = 0.5 * fov * np.pi / 180; inv = 2.0 / h; cx, cy = 0.5 * w, 0.5 * h
i = np.arange(h * w); row = i // w; col = i - row * w
u = (col + 0.5 - cx) * inv; v = (row + 0.5 - cy) * inv
rho = np.sqrt(u * u + v * v); th = rho * half_fov
ray = np.stack([np.sin(th) * u / rho, -np.sin(th) * v / rho, -np.cos(th)], 1)
(eseg, (pos, euler)) = list(EYE_CAMERAS.items())[0]
Rs = so.quat_to_mat(xquat[names.index(eseg)]); cam = xpos[names.index(eseg)] + Rs @ np.asarray(pos)
R = Rs @ so.quat_to_mat(_euler_xyz_extrinsic_quat(euler))
rw = ray @ R.T
r16 = rw.reshape(-1, 16, 3)
ax = r16.sum(1); ax /= np.linalg.norm(ax, axis=1)[:, None]
ccos = np.clip((r16 * ax[:, None, :]).sum(2).min(1) - 1e-5, -1, 1); alpha = np.arccos(ccos)
tot_old = tot_new = tot_exact = 0
tot_old_nw = tot_new_nw = 0
ch = np.arange(h * w // 16); nw = (ch * 16) // w == (ch * 16 + 15) // w
big = []
for c in range(len(seg)):
    Rc = so.quat_to_mat(xquat[seg[c]]); pa = xpos[seg[c]] + Rc @ geom[c, :3] - cam; pb = xpos[seg[c]] + Rc @ geom[c, 3:6] - cam; r = geom[c, 6]
    ba = pb - pa; L = np.linalg.norm(ba)
    # exact-ish: some ray passes within r of the segment (point-segment distance along the ray, sampled)
    # old: one disc
    mid = 0.5 * (pa + pb); d = np.linalg.norm(mid); Rb = 0.5 * L + r
    beta = np.arcsin(Rb / d) + 0.03 if d > Rb else 3.2
    ang = np.arccos(np.clip(ax @ (mid / d), -1, 1))
    old = (ang <= alpha + beta)
    new = np.zeros(len(ax), bool); betas = []
    for k in range(3):
        mk = pa + (2 * k + 1) / 6 * ba; dk = np.linalg.norm(mk); Rk = L / 6 + r
        bk = np.arcsin(Rk / dk) + 0.004 if dk > Rk else 3.2
        betas.append(bk)
        new |= np.arccos(np.clip(ax @ (mk / dk), -1, 1)) <= alpha + bk
    tot_old += old.sum(); tot_new += new.sum(); tot_old_nw += old[nw].sum(); tot_new_nw += new[nw].sum()
    big.append((names[seg[c]], round(float(np.degrees(beta)), 1), [round(float(np.degrees(b)), 1) for b in betas], int(old.sum()), int(new.sum())))
print("per-chunk candidates: one disc", tot_old / len(ax), "three discs", tot_new / len(ax))
print("without the chunks that wrap to the next row:", tot_old_nw / nw.sum(), tot_new_nw / nw.sum())
for b in sorted(big, key=lambda t: -t[4])[:12]: print(b)
