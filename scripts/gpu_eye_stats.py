#!/usr/bin/env python
"""Diagnostic (GPU box): culling statistics of the eye renderer on the bench's config-3 scene.  Needs a library built with
-DNMF_EYE_STATS (scripts/build_variant.sh eyestat -DNMF_TOPO_MASK=1 -DNMF_EYE_STATS) named by NMF_HIP_LIB."""
import ctypes, os, sys
from pathlib import Path
ROOT = Path(os.environ.get("GRAFT_REPO_ROOT") or Path(__file__).resolve().parents[1])
sys.path.insert(0, str(ROOT))
import torch
import bench
from flygym_amd import _native
args = bench.parse_args(["--vision", "render", "--steps", "40", "--warmup", "20", "--no-other-configs", "--no-cpu-baseline", "--no-live-counters"])
lib = _native.lib()
z = (ctypes.c_ulonglong * 8)()
bench.run(args)
lib.nmf_debug_eye_stats.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
lib.nmf_debug_eye_stats(z)
g = list(z)
groups = g[0]
print("groups", groups, "views", groups and "-")
print("capsules passing the group cull per group: %.2f" % (g[1] / max(groups, 1)))
print("capsules some chunk of the group keeps per group: %.2f" % (g[2] / max(groups, 1)))
print("(chunk, capsule) candidates per group: %.2f  (per chunk %.3f)" % (g[3] / max(groups, 1), g[3] / max(groups, 1) / 64))
print("groups with ground %.3f, spheres per group %.3f, groups with a capsule candidate %.3f" % (g[4] / max(groups, 1), g[5] / max(groups, 1), g[6] / max(groups, 1)))
print("sky-only groups %.3f, groups without ground and sphere %.3f" % ((g[7] & 0xffffffff) / max(groups, 1), (g[7] >> 32) / max(groups, 1)))
