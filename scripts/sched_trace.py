"""Diagnostic (gpurun, library built with -DNMF_SCHED_TRACE): where does a short chunked launch lose time?  Per workgroup of
one launch: start / exit times, items, cycles stepping vs between items."""
import ctypes, os, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import numpy as np, torch
from flygym_amd import HIPSimulation, make_model, _native
from flygym_amd.compose import ActuatorType
from flygym_amd.controllers import TripodCPG
n, spl = 4096, int(os.environ.get("SPL", "20"))
fly, world, _ = make_model()
order = fly.get_actuated_jointdofs_order(ActuatorType.POSITION)
sim = HIPSimulation(world, n_worlds=n, device=0)
table = TripodCPG(order, sim.timestep).targets(n, 2500, device=sim.device)
ids = sim.replay_ids(fly.name)
sim.set_leg_adhesion_states(fly.name, np.ones((n, 6), dtype=np.float32))
sim.step(500)
cur = 0
for _ in range(17):
    sim.step_replay(table, ids, cur, 50); cur += 50
L = _native.lib()
L.nmf_debug_sched_trace.argtypes = [ctypes.c_void_p, ctypes.c_int]
for k in range(6):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    sim.shader_clock_hz(reset=True)
    e0.record(); sim.step_replay(table, ids, cur, spl); e1.record(); cur += spl
    torch.cuda.synchronize()
    clk = sim.shader_clock_hz()
    buf = np.zeros((2048, 8), dtype=np.uint64)
    assert L.nmf_debug_sched_trace(buf.ctypes.data, 2048) == 0
    if k < 3:
        continue
    t0, t1 = buf[:, 0].astype(np.float64), buf[:, 1].astype(np.float64)
    start, end = t0.min(), t1.max()
    span_us = (end - start) / 100.0
    busy_us, gap_us = buf[:, 3] / clk * 1e6, buf[:, 4] / clk * 1e6
    print(f"launch {k}: event {e0.elapsed_time(e1) * 1e3:.0f} us, first start -> last exit {span_us:.0f} us; starts spread {(t0.max() - start) / 100:.1f} us; "
          f"exit: first {(t1.min() - start) / 100:.0f} median {(np.median(t1) - start) / 100:.0f} last {span_us:.0f} us (idle at the end, mean {np.mean(end - t1) / 100:.1f} us); "
          f"items per group mean {buf[:, 2].mean():.2f} (min {buf[:, 2].min()}, max {buf[:, 2].max()}); stepping {busy_us.mean():.0f} us, between items {gap_us.mean():.0f} us "
          f"= {gap_us.mean() / max(buf[:, 2].mean(), 1):.1f} us per item (state out {buf[:, 5].mean() / clk * 1e6 / 10:.2f}, ticket {buf[:, 6].mean() / clk * 1e6 / 10:.2f}, order {buf[:, 7].mean() / clk * 1e6 / 10:.2f} us); clock {clk / 1e9:.2f} GHz")
