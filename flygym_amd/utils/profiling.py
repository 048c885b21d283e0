"""Performance reports (reference ``src/flygym/utils/profiling.py:118-241``: ``print_perf_report`` for one world,
``print_perf_report_parallel`` for a batch).  Own layout; the quantities are the reference's: time per step / frame,
iterations per second, real-time factor; for a batch also the throughput over all worlds."""

from __future__ import annotations

__all__ = ["print_perf_report", "print_perf_report_parallel"]


def _rows(total_physics_time_ns, total_render_time_ns, n_steps, n_frames_rendered, timestep):
    if n_steps <= 0:
        raise ValueError("n_steps must be positive to compute a performance report")
    phys_us = total_physics_time_ns / 1e3 / n_steps
    rows = [("Physics", phys_us, 1e6 / phys_us if phys_us > 0 else float("nan"))]
    if n_frames_rendered > 0:
        rend_us = total_render_time_ns / 1e3 / n_frames_rendered
        rows.append(("Rendering", rend_us, 1e6 / rend_us if rend_us > 0 else float("nan")))
    total_us = (total_physics_time_ns + total_render_time_ns) / 1e3 / n_steps
    rows.append(("Total", total_us, 1e6 / total_us if total_us > 0 else float("nan")))
    return rows


def print_perf_report(total_physics_time_ns, total_render_time_ns, n_steps, n_frames_rendered, timestep) -> None:
    rows = _rows(total_physics_time_ns, total_render_time_ns, n_steps, n_frames_rendered, timestep)
    print("PERFORMANCE REPORT")
    print(f"{'stage':<12}{'us / iteration':>16}{'iterations / s':>18}{'x real time':>14}")
    for name, us, rate in rows:
        per = "frame" if name == "Rendering" else "step"
        rt = rate * timestep if name != "Rendering" else float("nan")
        print(f"{name:<12}{us:>16.2f}{rate:>18.1f}{rt:>14.4f}   (per {per})")
    if n_frames_rendered > 0:
        print(f"{n_frames_rendered} frames rendered over {n_steps} steps")
    else:
        print("No frames were rendered.")


def print_perf_report_parallel(total_physics_time_ns, total_render_time_ns, n_steps, n_frames_rendered, timestep,
                               n_worlds, n_worlds_rendered) -> None:
    rows = _rows(total_physics_time_ns, total_render_time_ns, n_steps, n_frames_rendered, timestep)
    print("PERFORMANCE REPORT")
    print(f"{'stage':<12}{'us / iteration':>16}{'iterations / s':>18}{'x real time':>14}{'parallelized throughput / s':>30}")
    for name, us, rate in rows:
        width = n_worlds_rendered if name == "Rendering" else n_worlds
        rt = rate * timestep if name != "Rendering" else float("nan")
        print(f"{name:<12}{us:>16.2f}{rate:>18.1f}{rt:>14.4f}{rate * width:>30.1f}")
    print(f"{n_worlds} worlds stepped in parallel" + (f", {n_worlds_rendered} rendered" if n_worlds_rendered else ""))
    if n_frames_rendered > 0:
        print(f"{n_frames_rendered} frames rendered over {n_steps} steps")
    else:
        print("No frames were rendered.")
