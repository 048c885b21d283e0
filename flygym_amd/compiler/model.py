"""Flat compiled model: the constants the stepping kernels consume.

The reference builds an MJCF document and lets MuJoCo's compiler turn it into an
``MjModel`` (``src/flygym/compose/base.py:21-27``).  Here the same inputs — rigging
table, meshes, joint/actuator/contact settings chosen through the ``Fly`` / ``World``
API — are compiled by :func:`compile_world` into a dictionary of float64 / int32
arrays (a :class:`CompiledModel`) that serialises to one binary blob read by both the
C oracle (``oracle/nmf_oracle.c``) and the HIP library (``flygym_amd/csrc``).

Compile-time semantics restated from MuJoCo's documented compiler behaviour
(parameters cited to the reference):

* ``boundmass=1e-6`` / ``boundinertia=1e-12`` lower clamps, ``angle=radian``
  (``assets/model/mujoco_globals.yaml:1-7``);
* every body has one geom with explicit ``mass`` (``fly.py:603-611``): inertia is the
  shape's unit-density inertia scaled to that mass;
* bodies without joints are rigidly merged into their moving ancestor for the
  dynamics ("static fusing"); the 69 named segment poses are kept as constant
  offsets so the observation surface is unchanged;
* ``body_invweight0`` (used by the contact regulariser) is evaluated per *named
  segment* at ``qpos0`` (all hinge angles 0, root at the spawn pose).
"""

from __future__ import annotations

import hashlib
import struct
from pathlib import Path

import numpy as np

from . import rigid
from .mesh import (MeshData, capsule_from_aabb, capsule_from_inertia_box, capsule_inertia, convex_mesh_data,
                   mirror_y)

__all__ = ["CompiledModel", "compile_world", "ASSET_PACK", "EngineSemantics"]

ASSET_PACK = Path(__file__).resolve().parents[1] / "assets" / "nmf_assets.npz"

GEOM_CAPSULE, GEOM_HULL = 0, 1
ACT_POSITION, ACT_ADHESION, ACT_MOTOR = 0, 1, 2

_MAGIC = b"NMFMODEL"
_VERSION = 4      # 4: sem_options[4] = terrain side faces; sem_options / act_geom mandatory (a v3 blob is refused, not half-read)


class EngineSemantics:
    """Named switches for the engine-stage semantics the reference leaves to MuJoCo and that could not be checked here
    (confidence "L"/"M" rows).  Each one is read by the model compiler, the C oracle and the HIP
    kernel alike, so a mismatch found with ``tests/golden/make_mujoco_golden.py`` on a box that has MuJoCo is a flag
    flip on ``world.semantics``, not a rewrite.  Defaults = this build's reading of MuJoCo 3.6's documentation.

    * ``mesh_inertia``: ``"exact"`` (signed tetrahedra over the triangle soup) | ``"convex"`` (inertia of the convex hull);
    * ``capsule_fit``: ``"inertia_box"`` (``fitaabb=false``: equivalent-inertia box) | ``"aabb"`` (``fitaabb=true``);
    * ``invweight0``: ``"segment"`` (per named segment at its own COM) | ``"fused_body"`` (per dynamic body after
      ``fusestatic``, at the merged COM — ADVICE r1);
    * ``pyramid_R``: ``"2mu2"`` (pyramidal rows regularised by ``2 mu^2 R_n``) | ``"plain"`` (``R_n`` on every edge);
    * ``adhesion_contacts``: ``"segment_geom"`` (contacts of the adhesion segment's own geom: the MJCF body the actuator
      names, ``fly.py:434-439``) | ``"fused_body"`` (every contact of the dynamic body it was fused into);
    * ``sensor_frame``: ``"world"`` (net force / torque in world axes) | ``"contact"`` (in the contact frame: normal,
      tangent 1, tangent 2 — the reference docstring's wording, ``simulation.py:226-231``);
    * ``max_hull_contacts``: 1..4 manifold points per plane-hull pair (4 = deepest vertex + up to 3 more within
      ``hull_skin``; 1 = deepest vertex only);
    * ``weld_relpose``: ``"spawn"`` (tether target = the root body's spawn pose) | ``"identity"`` (target = world origin,
      identity orientation);
    * ``terrain_walls``: ``"box"`` (the cells of a terrain are boxes: their side faces collide, horizontal normals —
      ``compose.world.terrain_probe``) | ``"heightfield"`` (round 2: tops only, vertical normals).  Build-defined terrains,
      no MuJoCo counterpart in the reference snapshot.
    """

    _CHOICES = dict(mesh_inertia=("exact", "convex"), capsule_fit=("inertia_box", "aabb"),
                    invweight0=("segment", "fused_body"), pyramid_R=("2mu2", "plain"),
                    adhesion_contacts=("segment_geom", "fused_body"), sensor_frame=("world", "contact"),
                    weld_relpose=("spawn", "identity"), terrain_walls=("box", "heightfield"))

    def __init__(self, **kw):
        self.mesh_inertia = "exact"
        self.capsule_fit = "inertia_box"
        self.invweight0 = "segment"
        self.pyramid_R = "2mu2"
        self.adhesion_contacts = "segment_geom"
        self.sensor_frame = "world"
        self.max_hull_contacts = 4
        self.weld_relpose = "spawn"
        self.terrain_walls = "box"
        for k, v in kw.items():
            if not hasattr(self, k):
                raise TypeError(f"unknown engine semantic '{k}'")
            setattr(self, k, v)
        self.validate()

    def validate(self):
        for k, choices in self._CHOICES.items():
            if getattr(self, k) not in choices:
                raise ValueError(f"engine semantic {k} must be one of {choices}, got {getattr(self, k)!r}")
        if int(self.max_hull_contacts) not in (1, 2, 3, 4):
            raise ValueError("max_hull_contacts must be 1..4")

    def flags(self) -> np.ndarray:
        """int32[8] blob entry ``sem_options`` read by the oracle and the kernel: pyramid_R plain, adhesion over the fused
        body, sensor in the contact frame, max hull contacts, terrain side faces collide, 0, 0, 0."""
        self.validate()
        return np.array([int(self.pyramid_R == "plain"), int(self.adhesion_contacts == "fused_body"),
                         int(self.sensor_frame == "contact"), int(self.max_hull_contacts), int(self.terrain_walls == "box"),
                         0, 0, 0], dtype=np.int32)

    def as_dict(self):
        self.validate()
        return {k: getattr(self, k) for k in (*self._CHOICES, "max_hull_contacts")}
_DT = {np.dtype(np.float64): 0, np.dtype(np.int32): 1}


class CompiledData:
    """Initial state of a compiled world (the counterpart of the ``mj_data`` the reference's ``compile()`` returns): the
    "neutral" keyframe.  The engine keeps the live state of every world on the GPU (``HIPSimulation.field``)."""

    def __init__(self, model: "CompiledModel"):
        self.qpos = np.array(model["key_qpos"], dtype=np.float64)
        self.qvel = np.zeros(model.nv)
        self.ctrl = np.array(model["key_ctrl"], dtype=np.float64)
        self.time = 0.0


class CompiledModel(dict):
    """``dict[str, np.ndarray]`` with (de)serialisation; keys are the blob entry names."""

    meta: dict

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self.meta = {}

    # sizes ---------------------------------------------------------------
    @property
    def nb(self): return int(self["body_parent"].shape[0])
    @property
    def nv(self): return int(self["dof_body"].shape[0])
    @property
    def nq(self): return self.nv + 1
    @property
    def nu(self): return int(self["act_type"].shape[0])
    @property
    def ng(self): return int(self["geom_body"].shape[0])
    @property
    def nseg(self): return int(self["seg_body"].shape[0])
    @property
    def nsite(self): return int(self["site_body"].shape[0])
    # MuJoCo's names for the same counts (what reference code reads off ``mj_model``)
    @property
    def nbody(self): return self.nseg + 1                       # named segments + the world body
    @property
    def njnt(self): return self.nv - 6 + 1                      # hinges + the free joint
    @property
    def ngeom(self): return self.ng + 1                         # contact geoms + the ground
    @property
    def ncam(self): return int(self.meta.get("n_cameras", 0))

    # blob ----------------------------------------------------------------
    def to_blob(self) -> bytes:
        names = sorted(self.keys())
        head = struct.calcsize("<8sII")
        ent = struct.calcsize("<32sII4qqq")
        off = head + ent * len(names)
        off = (off + 63) // 64 * 64
        table, chunks = [], []
        for n in names:
            a = np.ascontiguousarray(self[n])
            if a.dtype not in _DT:
                a = a.astype(np.float64 if a.dtype.kind == "f" else np.int32)
            shape = list(a.shape) + [0] * (4 - a.ndim)
            table.append(struct.pack("<32sII4qqq", n.encode(), _DT[a.dtype], a.ndim, *shape, off, a.nbytes))
            chunks.append((off, a.tobytes()))
            off = (off + a.nbytes + 63) // 64 * 64
        buf = bytearray(off)
        buf[0:head] = struct.pack("<8sII", _MAGIC, _VERSION, len(names))
        for i, t in enumerate(table):
            buf[head + i * ent: head + (i + 1) * ent] = t
        for o, b in chunks:
            buf[o:o + len(b)] = b
        return bytes(buf)

    @classmethod
    def from_blob(cls, blob: bytes) -> "CompiledModel":
        magic, ver, n = struct.unpack_from("<8sII", blob, 0)
        if magic != _MAGIC or ver != _VERSION:
            raise ValueError("not an NMFMODEL v%d blob" % _VERSION)
        head = struct.calcsize("<8sII")
        ent = struct.calcsize("<32sII4qqq")
        out = cls()
        for i in range(n):
            name, dt, nd, s0, s1, s2, s3, off, nbytes = struct.unpack_from("<32sII4qqq", blob, head + i * ent)
            dtype = np.float64 if dt == 0 else np.int32
            shape = (s0, s1, s2, s3)[:nd]
            out[name.rstrip(b"\0").decode()] = np.frombuffer(blob, dtype=dtype, count=int(np.prod(shape, dtype=np.int64)), offset=off).reshape(shape).copy()
        return out

    def save(self, path):
        Path(path).write_bytes(self.to_blob())

    @classmethod
    def load(cls, path):
        return cls.from_blob(Path(path).read_bytes())

    def digest(self) -> str:
        return hashlib.sha256(self.to_blob()).hexdigest()[:16]


# ----------------------------------------------------------------------------
# asset access
# ----------------------------------------------------------------------------
_pack_cache = {}


def load_asset_pack(path=None):
    path = Path(path or ASSET_PACK)
    if path not in _pack_cache:
        if not path.exists():
            raise FileNotFoundError(
                f"asset pack {path} missing: run scripts/build_asset_pack.py against a flygym asset directory"
            )
        _pack_cache[path] = dict(np.load(path, allow_pickle=False))
    return _pack_cache[path]


def mesh_for_segment(pack, seg_name: str, mesh_type: str, mirror_left2right=True) -> MeshData:
    """Mesh constants for a segment, with the reference's fallback and mirroring rules
    (``fly.py:514-543``): right-side segments reuse the left mesh under y → −y; a mesh
    absent from ``mesh_type`` falls back to ``fullsize``."""
    src = seg_name
    flip = False
    if mirror_left2right and seg_name[0] == "r":
        src, flip = "l" + seg_name[1:], True
    for mt in (mesh_type, "fullsize"):
        key = f"mesh/{mt}/{src}"
        if key + "/props" in pack:
            p = pack[key + "/props"]
            md = MeshData(
                volume=float(p[0]), hull_volume=float(p[1]), n_vertices=int(p[2]), n_faces=int(p[3]),
                com=p[4:7].copy(), inertia=p[7:16].reshape(3, 3).copy(),
                hull_vertices=pack[key + "/hull_v"].copy(), hull_faces=pack[key + "/hull_f"].copy(),
            )
            return mirror_y(md) if flip else md
    raise FileNotFoundError(f"Mesh file not found for segment {seg_name}")


# ----------------------------------------------------------------------------
# compile
# ----------------------------------------------------------------------------
def _segment_inertial(md: MeshData, mass: float, as_capsule: bool, boundinertia: float, sem: "EngineSemantics"):
    """(ipos, principal rotation, principal moments, capsule(r, half, centre)|None) in segment frame."""
    if sem.mesh_inertia == "convex":
        md = convex_mesh_data(md)
    w, R = md.principal()
    cap = None
    ipos = md.com.copy()
    if as_capsule:
        if sem.capsule_fit == "aabb":
            r, half, centre = capsule_from_aabb(md)
            ipos = centre          # the fitted primitive (and with it the body's inertial frame) sits at the box centre
        else:
            r, half = capsule_from_inertia_box(md)
            centre = md.com.copy()
        moments = capsule_inertia(r, half, mass)
        cap = (r, half, centre)
    else:
        moments = mass * w / md.volume
    moments = np.maximum(moments, boundinertia)
    return ipos, R, moments, cap


def compile_world(world) -> CompiledModel:
    """Compile a ``flygym_amd.compose`` world holding exactly one fly."""
    if len(world.fly_lookup) != 1:
        raise ValueError("the MI355X engine steps exactly one fly per world (batch = many worlds)")
    fly = next(iter(world.fly_lookup.values()))
    pack = load_asset_pack(fly.asset_pack_path)
    opt = fly.mujoco_globals
    sem = getattr(world, "semantics", None) or EngineSemantics()
    sem.validate()
    boundmass = float(opt["compiler"].get("boundmass", 0.0))
    boundinertia = float(opt["compiler"].get("boundinertia", 0.0))

    seg_names = [s.name for s in fly.get_bodysegs_order()]
    seg_index = {n: i for i, n in enumerate(seg_names)}
    ns = len(seg_names)
    rig_idx = {str(n): i for i, n in enumerate(pack["rigging_names"])}

    seg_parent = np.full(ns, -1, dtype=np.int64)
    for parent, child in fly.segment_edges():
        seg_parent[seg_index[child]] = seg_index[parent]

    seg_pos = np.zeros((ns, 3))
    seg_quat = np.zeros((ns, 4))
    seg_mass = np.zeros(ns)
    seg_ipos = np.zeros((ns, 3))
    seg_imat = np.zeros((ns, 3, 3))        # inertia about COM, segment frame
    seg_geomR = np.zeros((ns, 3, 3))       # geom (principal) frame in segment frame
    seg_capsule = [None] * ns
    seg_mesh = [None] * ns
    for i, n in enumerate(seg_names):
        r = rig_idx[n]
        seg_pos[i] = pack["rigging_pos"][r]
        seg_quat[i] = rigid.quat_normalize(pack["rigging_quat"][r])
        seg_mass[i] = max(float(pack["rigging_mass"][r]), boundmass)
        md = mesh_for_segment(pack, n, fly.mesh_type.value, fly.mirror_left2right)
        ipos, R, moments, cap = _segment_inertial(md, seg_mass[i], fly.segment_is_capsule(n), boundinertia, sem)
        seg_ipos[i], seg_geomR[i] = ipos, R
        seg_imat[i] = R @ np.diag(moments) @ R.T
        seg_capsule[i], seg_mesh[i] = cap, md

    # ---- dofs per segment, in skeleton order --------------------------------
    jointdofs = fly.get_jointdofs_order()
    seg_dofs: dict[int, list] = {}
    for d in jointdofs:
        seg_dofs.setdefault(seg_index[d.child.name], []).append(d)

    # ---- dynamic bodies: attach body (free joint) + every segment with hinges --
    # body 0 is the attachment frame created by spawn_site.attach(...).add("freejoint")
    # (world.py:274-277); the root segment hangs off it at its rigging pose.
    dyn_of_seg = np.full(ns, -1, dtype=np.int64)
    seg_off_pos = np.zeros((ns, 3))             # pose of the segment frame in its dyn body frame
    seg_off_mat = np.zeros((ns, 3, 3))
    body_parent, body_pos, body_quat, body_seg = [-1], [np.zeros(3)], [np.array([1.0, 0, 0, 0])], [-1]
    order = list(range(ns))  # seg order is already parent-before-child (DFS)
    for s in order:
        p = seg_parent[s]
        Rs = rigid.quat_to_mat(seg_quat[s])
        if p < 0:
            par_body, par_pos, par_mat = 0, np.zeros(3), np.eye(3)
        else:
            par_body, par_pos, par_mat = dyn_of_seg[p], seg_off_pos[p], seg_off_mat[p]
        pos_in_parbody = par_pos + par_mat @ seg_pos[s]
        mat_in_parbody = par_mat @ Rs
        if s in seg_dofs:
            dyn_of_seg[s] = len(body_parent)
            body_parent.append(int(par_body))
            body_pos.append(pos_in_parbody)
            body_quat.append(rigid.mat_to_quat(mat_in_parbody))
            body_seg.append(s)
            seg_off_pos[s], seg_off_mat[s] = np.zeros(3), np.eye(3)
        else:
            dyn_of_seg[s] = par_body
            seg_off_pos[s], seg_off_mat[s] = pos_in_parbody, mat_in_parbody
    nb = len(body_parent)

    # ---- fused inertias ----------------------------------------------------
    body_mass = np.zeros(nb)
    body_mc = np.zeros((nb, 3))
    body_I0 = np.zeros((nb, 3, 3))  # about the dyn body origin
    # the attachment body itself: massless in MJCF → boundmass / boundinertia
    body_mass[0] += boundmass
    body_I0[0] += np.eye(3) * boundinertia
    for s in range(ns):
        b = dyn_of_seg[s]
        c = seg_off_pos[s] + seg_off_mat[s] @ seg_ipos[s]
        Ic = seg_off_mat[s] @ seg_imat[s] @ seg_off_mat[s].T
        m = seg_mass[s]
        body_mass[b] += m
        body_mc[b] += m * c
        body_I0[b] += Ic + m * (np.dot(c, c) * np.eye(3) - np.outer(c, c))
    body_ipos = body_mc / body_mass[:, None]
    body_inertia = np.zeros((nb, 6))
    for b in range(nb):
        c = body_ipos[b]
        Ic = body_I0[b] - body_mass[b] * (np.dot(c, c) * np.eye(3) - np.outer(c, c))
        body_inertia[b] = rigid.mat_to_sym6(Ic)

    # ---- dof tables ----------------------------------------------------------
    nv = 6 + len(jointdofs)
    dof_body = np.zeros(nv, dtype=np.int32)
    dof_parent = np.full(nv, -1, dtype=np.int32)
    dof_axis = np.zeros((nv, 3))
    dof_armature = np.zeros(nv)
    dof_damping = np.zeros(nv)
    dof_stiffness = np.zeros(nv)
    dof_springref = np.zeros(nv)
    body_dofadr = np.zeros(nb, dtype=np.int32)
    body_dofnum = np.zeros(nb, dtype=np.int32)
    body_dofnum[0] = 6
    for i in range(6):
        dof_parent[i] = i - 1
    dof_axis[3:6] = np.eye(3)
    dof_index = {}
    last_dof_of_body = {0: 5}
    adr = 6
    for b in range(1, nb):
        s = body_seg[b]
        body_dofadr[b] = adr
        prev = last_dof_of_body[body_parent[b]]
        for d in seg_dofs[s]:
            jp = fly.joint_params[d]
            dof_body[adr] = b
            dof_parent[adr] = prev
            dof_axis[adr] = jp["axis"]
            dof_armature[adr] = jp["armature"]
            dof_damping[adr] = jp["damping"]
            dof_stiffness[adr] = jp["stiffness"]
            dof_springref[adr] = jp["springref"]
            dof_index[d] = adr
            prev = adr
            adr += 1
        body_dofnum[b] = adr - body_dofadr[b]
        last_dof_of_body[b] = prev
    # the compose layer iterates jointdofs in DFS order, bodies were created in the same
    # order, so dof addresses follow fly.get_jointdofs_order() exactly:
    assert [dof_index[d] for d in jointdofs] == list(range(6, nv))

    m = CompiledModel()
    m["opt_timestep"] = np.array([float(opt["option"]["timestep"])])
    m["opt_gravity"] = np.array(opt["option"]["gravity"], dtype=np.float64)
    m["opt_solver"] = np.array([int(opt["option"].get("iterations", 100)), int(world.noslip_iterations)], dtype=np.int32)
    m["opt_tolerance"] = np.array([1e-8])
    m["body_parent"] = np.array(body_parent, dtype=np.int32)
    m["body_pos"] = np.array(body_pos)
    m["body_quat"] = np.array(body_quat)
    m["body_mass"] = body_mass
    m["body_ipos"] = body_ipos
    m["body_inertia"] = body_inertia
    m["body_dofadr"] = body_dofadr
    m["body_dofnum"] = body_dofnum
    m["dof_body"] = dof_body
    m["dof_parent"] = dof_parent
    m["dof_axis"] = dof_axis
    m["dof_armature"] = dof_armature
    m["dof_damping"] = dof_damping
    m["dof_stiffness"] = dof_stiffness
    m["dof_springref"] = dof_springref
    m["seg_body"] = dyn_of_seg.astype(np.int32)
    m["seg_pos"] = seg_off_pos
    m["seg_quat"] = np.array([rigid.mat_to_quat(seg_off_mat[s]) for s in range(ns)])

    # ---- sites ---------------------------------------------------------------
    site_segs = [seg_index[j.child.name] for j in fly.get_sites_order()]
    m["site_body"] = np.array([dyn_of_seg[s] for s in site_segs], dtype=np.int32).reshape(-1)
    m["site_pos"] = np.array([seg_off_pos[s] for s in site_segs]).reshape(-1, 3)

    # ---- actuators (MJCF order: in the order they were added) -----------------
    act_type, act_trn, act_gain, act_bias, act_frc, act_ctrl, act_lim = [], [], [], [], [], [], []
    general_rows = {}          # actuator index -> its row of act_general (the types beyond the stateless affine ones)
    affine_dofs = set()        # dofs the kernel's affine pass already writes: one actuator per dof there (lane = actuator, plain stores)
    for a in fly.actuators:
        if a["kind"] != "adhesion" and a["kind"] not in _GENERAL_KINDS:
            if dof_index[a["jointdof"]] in affine_dofs:
                # a second actuator on a dof (the reference allows several types on one joint, compose/fly.py:310-312): the same
                # affine law as a row of the general pass, which adds to the dof's force atomically
                a = dict(a, kind="affine", jointdof_kind=a["kind"])
            else:
                affine_dofs.add(dof_index[a["jointdof"]])
        if a["kind"] in _GENERAL_KINDS or a["kind"] == "affine":
            # the stepping kernel's affine pass sees a motor of gain 0 without a force limit; the general pass (nmf_step.hip
            # actuation_general, oracle general_actuator) computes the force from the act_general row
            general_rows[len(act_type)] = _general_row(a, dof_index[a["jointdof"]])
            act_type.append(ACT_MOTOR)
            act_trn.append(dof_index[a["jointdof"]])
            act_gain.append(0.0)
            act_bias.append([0.0, 0.0])
            act_frc.append(a["forcerange"])
            act_ctrl.append(a["ctrlrange"])
            act_lim.append([0, int(a["ctrllimited"])])
            continue
        if a["kind"] == "adhesion":
            act_type.append(ACT_ADHESION)
            act_trn.append(int(dyn_of_seg[seg_index[a["segment"]]]))
            act_gain.append(a["gain"])
            act_bias.append([0.0, 0.0])
        else:
            act_type.append(ACT_POSITION if a["kind"] == "position" else ACT_MOTOR)
            act_trn.append(dof_index[a["jointdof"]])
            kp, kv = a.get("kp", 1.0), a.get("kv", 0.0)
            if a["kind"] == "position":
                act_gain.append(kp)
                act_bias.append([-kp, -kv])
            elif a["kind"] == "velocity":       # MuJoCo's velocity servo: kv (ctrl - qd); kv defaults to 1
                kv = a.get("kv", 1.0)
                act_gain.append(kv)
                act_bias.append([0.0, -kv])
            else:
                act_gain.append(a.get("gear", 1.0))
                act_bias.append([0.0, 0.0])
        act_frc.append(a["forcerange"])
        act_ctrl.append(a["ctrlrange"])
        act_lim.append([int(a["forcelimited"]), int(a["ctrllimited"])])
    nu = len(act_type)
    m["act_type"] = np.array(act_type, dtype=np.int32).reshape(nu)
    m["act_trn"] = np.array(act_trn, dtype=np.int32).reshape(nu)
    m["act_gain"] = np.array(act_gain, dtype=np.float64).reshape(nu)
    m["act_bias"] = np.array(act_bias, dtype=np.float64).reshape(nu, 2)
    m["act_forcerange"] = np.array(act_frc, dtype=np.float64).reshape(nu, 2)
    m["act_ctrlrange"] = np.array(act_ctrl, dtype=np.float64).reshape(nu, 2)
    m["act_limited"] = np.array(act_lim, dtype=np.int32).reshape(nu, 2)
    m["sem_options"] = sem.flags()

    # ---- keyframe "neutral" (fly.py:658-678, world.py:151-207) ----------------
    qpos = np.zeros(nv + 1)
    qpos[0:3] = world.spawn_position
    qpos[3:7] = rigid.quat_normalize(world.spawn_quat)
    for d, ang in fly.jointdof_to_neutralangle.items():
        qpos[dof_index[d] + 1] = ang
    m["key_qpos"] = qpos
    m["key_ctrl"] = np.array([a["neutral"] for a in fly.actuators], dtype=np.float64).reshape(nu)
    qpos0 = np.zeros(nv + 1)
    qpos0[0:7] = qpos[0:7]
    m["qpos0"] = qpos0

    # ---- contact geoms ---------------------------------------------------------
    cps = world.ground_contact_params
    contact_segs = [seg_index[s.name] for s in world.bodysegs_with_ground_contact]
    # geoms ordered by the dynamic body that carries them (stable: the preset's order within a body): the engine keeps
    # a fly's contacts sorted by body.  The leg / thorax / abdomen / head presets already are.
    contact_segs = sorted(contact_segs, key=lambda s: int(dyn_of_seg[s]))
    g_body, g_seg, g_type, g_p0, g_p1, g_rad, g_hadr, g_hnum, g_bs = [], [], [], [], [], [], [], [], []
    hull_chunks, hadr = [], 0
    for s in contact_segs:
        md = seg_mesh[s]
        T_pos, T_mat = seg_off_pos[s], seg_off_mat[s]
        g_body.append(int(dyn_of_seg[s]))
        g_seg.append(s)
        if seg_capsule[s] is not None:
            r, half, c = seg_capsule[s]
            zax = seg_geomR[s][:, 2]
            p0 = T_pos + T_mat @ (c - zax * half)
            p1 = T_pos + T_mat @ (c + zax * half)
            g_type.append(GEOM_CAPSULE)
            g_p0.append(p0); g_p1.append(p1); g_rad.append(r)
            g_hadr.append(0); g_hnum.append(0)
            g_bs.append([*(0.5 * (p0 + p1)), half + r])
        else:
            hv = (T_mat @ md.hull_vertices.T).T + T_pos
            g_type.append(GEOM_HULL)
            # bounding cylinder of the hull (broad phase only; the fields are unused for hulls otherwise): axis = first
            # principal direction, ends at the extreme projections, radius = largest distance from the axis, inflated a
            # little so the bound stays conservative in float32
            cen = hv.mean(axis=0)
            ax = np.linalg.svd(hv - cen, full_matrices=False)[2][0]
            tpar = (hv - cen) @ ax
            rad = np.linalg.norm((hv - cen) - np.outer(tpar, ax), axis=1).max()
            g_p0.append(cen + tpar.min() * ax); g_p1.append(cen + tpar.max() * ax); g_rad.append(float(rad * (1 + 1e-4) + 1e-6))
            g_hadr.append(hadr); g_hnum.append(len(hv))
            hull_chunks.append(hv)
            hadr += len(hv)
            centre = 0.5 * (hv.min(axis=0) + hv.max(axis=0))
            g_bs.append([*centre, float(np.linalg.norm(hv - centre, axis=1).max())])
    ng = len(contact_segs)
    m["geom_body"] = np.array(g_body, dtype=np.int32).reshape(ng)
    m["geom_seg"] = np.array(g_seg, dtype=np.int32).reshape(ng)
    m["geom_type"] = np.array(g_type, dtype=np.int32).reshape(ng)
    m["geom_p0"] = np.array(g_p0, dtype=np.float64).reshape(ng, 3)
    m["geom_p1"] = np.array(g_p1, dtype=np.float64).reshape(ng, 3)
    m["geom_radius"] = np.array(g_rad, dtype=np.float64).reshape(ng)
    m["geom_hulladr"] = np.array(g_hadr, dtype=np.int32).reshape(ng)
    m["geom_hullnum"] = np.array(g_hnum, dtype=np.int32).reshape(ng)
    m["geom_bsphere"] = np.array(g_bs, dtype=np.float64).reshape(ng, 4)
    m["hull_vert"] = np.concatenate(hull_chunks, axis=0) if hull_chunks else np.zeros((0, 3))
    fr = cps.get_friction_tuple()
    sr = cps.get_solref_tuple()
    si = _solimp5(cps.get_solimp_tuple())
    m["pair_friction"] = np.tile(np.array(fr, dtype=np.float64), (ng, 1)).reshape(ng, 5)
    m["pair_solref"] = np.tile(np.array(sr, dtype=np.float64), (ng, 1)).reshape(ng, 2)
    m["pair_solimp"] = np.tile(si, (ng, 1)).reshape(ng, 5)
    m["pair_margin"] = np.full(ng, float(cps.margin))
    m["hull_skin"] = np.array([1e-3])

    # leg contact sensors (world.py:311-331): geoms of a leg whose segment is at or
    # below the most proximal contacting segment feed that leg's sensor
    from ..anatomy import LEGS
    g_sensor = np.full(ng, -1, dtype=np.int32)
    if world.add_ground_contact_sensors:
        for gi, s in enumerate(contact_segs):
            pos = seg_names[s].split("_")[0]
            if pos in LEGS:
                g_sensor[gi] = LEGS.index(pos)
    m["geom_sensor"] = g_sensor
    # adhesion actuators act through the contacts of their own segment's geom (the MJCF body the actuator names,
    # reference fly.py:434-439; a body an actuator references is not fused away by MuJoCo's fusestatic), not through
    # every contact of the dynamic body the segment was merged into here.  -1: the segment has no contact geom.
    geom_of_seg = {s: gi for gi, s in enumerate(contact_segs)}
    m["act_geom"] = np.array([geom_of_seg.get(seg_index[a["segment"]], -1) if a["kind"] == "adhesion" else -1
                              for a in fly.actuators], dtype=np.int32).reshape(nu)
    m["n_sensor"] = np.array([6 if world.add_ground_contact_sensors else 0], dtype=np.int32)

    # adhesion: actuator index per leg order (simulation.py:387-404)
    m["plane"] = np.array([0.0, 0.0, 1.0, 0.0])  # ground plane n·x = d  (world.py:251-261)
    # height map on top of the plane (0 flat, 1 gapped, 2 blocks, 3 mixed) + its maximum height (cull bound)
    m["terrain_type"] = np.array([int(world.terrain_type)], dtype=np.int32)
    tp = np.array(world.terrain_params, dtype=np.float64)
    hmax = {0: 0.0, 1: 0.0, 2: tp[1], 3: 0.35}[int(world.terrain_type)]
    m["terrain_params"] = np.concatenate([tp, [hmax]])

    # ---- invweight0 at qpos0 ---------------------------------------------------
    M0 = rigid.mass_matrix_from_jacobians(m, qpos0)
    m["stat_meaninertia"] = np.array([float(np.mean(np.diag(M0)))])
    Minv = np.linalg.inv(M0)
    xpos, xmat, xquat = rigid.forward_kinematics(m, qpos0)
    axis, anchor = rigid.dof_axes_world(m, qpos0, xpos, xmat, xquat)
    seg_invw = np.zeros((ns, 2))
    for s in range(ns):
        b = int(dyn_of_seg[s])
        com = xpos[b] + xmat[b] @ (seg_off_pos[s] + seg_off_mat[s] @ seg_ipos[s])
        J = rigid.point_jacobian(m, b, com, axis, anchor)
        A = J @ Minv @ J.T
        seg_invw[s] = [np.trace(A[0:3, 0:3]) / 3.0, np.trace(A[3:6, 3:6]) / 3.0]
    if sem.invweight0 == "fused_body":
        # fusestatic: geoms of jointless segments take the merged body's invweight0, evaluated at the merged COM
        body_invw = np.zeros((nb, 2))
        for b in range(nb):
            com = xpos[b] + xmat[b] @ body_ipos[b]
            J = rigid.point_jacobian(m, b, com, axis, anchor)
            A = J @ Minv @ J.T
            body_invw[b] = [np.trace(A[0:3, 0:3]) / 3.0, np.trace(A[3:6, 3:6]) / 3.0]
        referenced = {seg_index[a["segment"]] for a in fly.actuators if a["kind"] == "adhesion"}
        for s in range(ns):
            if s not in seg_dofs and s not in referenced:
                seg_invw[s] = body_invw[int(dyn_of_seg[s])]
    m["seg_invweight0"] = seg_invw
    if general_rows:
        # acc0 of MuJoCo's actuators: |M^-1 moment| at qpos0 (the muscle model's peak force when `force` < 0 is scale / acc0)
        gen = np.zeros((nu, _ACTGEN))
        for u, row in general_rows.items():
            row[31] = abs(row[5]) * float(np.linalg.norm(Minv[:, int(m["act_trn"][u])]))
            gen[u] = row
        m["act_general"] = gen
    m["geom_invweight0"] = seg_invw[contact_segs, 0].reshape(ng) if ng else np.zeros(0)

    # tether weld (TetheredWorld): six bilateral rows holding the root body at its spawn pose.  The reference welds
    # the thorax with solref (2e-4, 1), solimp (0.98, 0.99, 1e-5, 0.5, 3) (world.py:358-365); the regulariser uses
    # the thorax body's invweight0 (translational, rotational).
    root_seg = seg_index[fly.root_segment.name]
    m["weld_active"] = np.array([1 if world.fixed_base else 0], dtype=np.int32)
    weld_target = np.concatenate([qpos[0:3], qpos[3:7]]) if sem.weld_relpose == "spawn" else np.array([0, 0, 0, 1.0, 0, 0, 0])
    m["weld_params"] = np.concatenate([weld_target, [2e-4, 1.0], _solimp5((0.98, 0.99, 1e-5, 0.5, 3.0)),
                                       seg_invw[root_seg]])
    # structure summary for the star-of-chains fast path
    m["star"] = _star_structure(m)
    m.meta = {
        "n_cameras": len(fly.cameraname_to_camera),
        "seg_names": seg_names,
        "dof_names": [d.name for d in jointdofs],
        "actuator_names": [a["name"] for a in fly.actuators],
        "fly_name": fly.name,
        "semantics": sem.as_dict(),
    }
    return m


_GENERAL_KINDS = ("intvelocity", "damper", "cylinder", "muscle")
_ACTGEN = 32


def _general_row(a: dict, dof: int) -> np.ndarray:
    """One row of ``act_general``: MuJoCo's general actuator (dyntype / gaintype / biastype with their parameter vectors) as the
    actuator shortcuts of the XML reference expand to it (``XMLreference.html#actuator-intvelocity`` ... ``-muscle``; reference
    ``compose/fly.py:65-77, 301-369`` forwards the shortcut and its attributes to MJCF).  Layout: 0 flags (1 on | 2
    forcelimited | the actuated dof << 8 — the kernel's general pass takes the dof from here: the copy of ``act_trn`` it uploads
    points these actuators at dof 0, so that the affine pass's plain store of their zero force races with nobody), 1 dyntype (0 none, 1 integrator, 2 filter, 3 filterexact, 4 muscle), 2 gaintype (0 fixed, 1 affine, 2 muscle),
    3 biastype (0 none, 1 affine, 2 muscle), 4 actlimited, 5 gear, 6..8 dynprm, 9..17 gainprm, 18..26 biasprm, 27..28 actrange,
    29..30 lengthrange, 31 acc0 (filled in by the caller)."""
    r = np.zeros(_ACTGEN)
    kind = a["kind"]
    r[0] = 1 + (2 if a["forcelimited"] else 0) + (int(dof) << 8)
    r[5] = a.get("gear", 1.0)
    if kind == "affine":                # position / velocity / motor on a dof that another actuator already drives (see compile_world)
        r[5] = 1.0                      # (a motor's gear is folded into its gain, as in the affine pass)
        r[1], r[2], r[3] = 0, 0, 1
        kp, kv = a.get("kp", 1.0), a.get("kv", 0.0)
        if a["jointdof_kind"] == "position":
            r[9], r[18:21] = kp, (0.0, -kp, -kv)
        elif a["jointdof_kind"] == "velocity":
            kv = a.get("kv", 1.0)
            r[9], r[18:21] = kv, (0.0, 0.0, -kv)
        else:
            r[9] = a.get("gear", 1.0)
    elif kind == "intvelocity":           # integrator; force = kp (act - q) - kv qd; the activation is clamped to actrange
        kp, kv = a.get("kp", 1.0), a.get("kv", 0.0)
        r[1], r[2], r[3], r[4] = 1, 0, 1, 1
        r[9] = kp
        r[18:21] = (0.0, -kp, -kv)
        r[27:29] = a["actrange"]
    elif kind == "damper":              # force = -kv * velocity * ctrl (affine gain, no bias), ctrl >= 0
        r[1], r[2], r[3] = 0, 1, 0
        r[9:12] = (0.0, 0.0, -a.get("kv", 1.0))
    elif kind == "cylinder":            # first-order filter of the control; force = area * act + bias . (1, length, velocity)
        r[1], r[2], r[3] = 2, 0, 1
        r[6] = a.get("timeconst", 1.0)
        r[9] = a.get("area", 1.0)
        r[18:21] = a.get("bias", (0.0, 0.0, 0.0))
    elif kind == "muscle":
        r[1], r[2], r[3] = 4, 2, 2
        r[6:8] = a.get("timeconst", (0.01, 0.04))
        r[8] = a.get("tausmooth", 0.0)
        prm = (*a.get("range", (0.75, 1.05)), a.get("force", -1.0), a.get("scale", 200.0), a.get("lmin", 0.5), a.get("lmax", 1.6),
               a.get("vmax", 1.5), a.get("fpmax", 1.3), a.get("fvmax", 1.2))
        r[9:18] = prm
        r[18:27] = prm
        r[29:31] = a["lengthrange"]
    else:
        raise ValueError(kind)
    return r


def _solimp5(t):
    """MJCF fills the values given and keeps the defaults for the rest
    (default solimp = 0.9 0.95 0.001 0.5 2); the reference passes FOUR numbers
    (``compose/physics.py:103-111``) so its midpoint/sharpness land in width/midpoint.
    The engine then clamps: d0,dmax,midpoint to [1e-4, 0.9999], width ≥ 0, power ≥ 1."""
    full = np.array([0.9, 0.95, 0.001, 0.5, 2.0])
    full[: len(t)] = t
    lo, hi = 1e-4, 0.9999
    full[0] = min(max(full[0], lo), hi)
    full[1] = min(max(full[1], lo), hi)
    full[2] = max(full[2], 0.0)
    full[3] = min(max(full[3], lo), hi)
    full[4] = max(full[4], 1.0)
    return full


def _star_structure(m) -> np.ndarray:
    """[is_star, n_chain, chain_dofs, chain_bodies] — 1 if every non-root body lies on one of
    ``n_chain`` equal serial chains hanging off the root (the HIP fast path's layout)."""
    nb = m["body_parent"].shape[0]
    parent = m["body_parent"]
    children = {b: [] for b in range(nb)}
    for b in range(1, nb):
        children[int(parent[b])].append(b)
    chains = []
    ok = True
    for c in children[0]:
        chain = [c]
        while len(children[chain[-1]]) == 1:
            chain.append(children[chain[-1]][0])
        if children[chain[-1]]:
            ok = False
        chains.append(chain)
    lens = {len(c) for c in chains}
    dofs = {int(sum(m["body_dofnum"][b] for b in c)) for c in chains}
    contiguous = all(c == list(range(c[0], c[0] + len(c))) for c in chains)
    if ok and len(lens) == 1 and len(dofs) == 1 and contiguous and chains:
        return np.array([1, len(chains), dofs.pop(), lens.pop()], dtype=np.int32)
    return np.array([0, len(chains), 0, 0], dtype=np.int32)
