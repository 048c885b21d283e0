"""Offline model of a chunked launch (diagnostic): list scheduling of (world, steps) tickets on 2048 persistent workgroups with
the precedence of a world's items, on per-world costs recorded by scripts/archive/cost_pairing.py (gpurun_out/costpair_chunks.npy).
Reproduces the measured failure of whole-launch items for the costliest worlds (2275 vs 1894 us modelled, 38.1 vs 44.7 M measured)."""
import numpy as np, heapq, sys
c = np.load("gpurun_out/costpair_chunks.npy")[0]
clk = 2.39e3
def sim(true, tickets, over=7.2):
    """tickets: list of (world, steps) in pull order; precedence: a world's items in sequence."""
    per_step = true/20/clk
    heap = [0.0]*2048; heapq.heapify(heap)
    ready = {}
    for w, steps in tickets:
        t = heapq.heappop(heap)
        start = max(t, ready.get(w, 0.0))
        end = start + steps*per_step[w] + over
        ready[w] = end
        heapq.heappush(heap, end)
    return max(heap)
def plan_tickets(order, K, planA, planB):
    tk = []
    A, B = order[:K], order[K:]
    # tickets: A's first chunk, then rounds
    if planA == "whole":
        tk += [(w, 20) for w in A]
        for s in planB: tk += [(w, s) for w in B]
    else:
        nr = max(len(planA), len(planB))
        for r in range(nr):
            if r < len(planA): tk += [(w, planA[r]) for w in A]
            if r < len(planB): tk += [(w, planB[r]) for w in B]
    return tk
std = [10,5,3,1,1]
res = {}
for name, K, pA, pB in [("std", 0, std, std), ("whole2048", 2048, "whole", std), ("whole512", 512, "whole", std),
                        ("A:10,10 K1024", 1024, [10,10], std), ("A:10,10 K2048", 2048, [10,10], std), ("A:10,6,4 K2048", 2048, [10,6,4], std),
                        ("A:12,8 K1024", 1024, [12,8], std), ("A:10,5,5 K2048", 2048, [10,5,5], std), ("A:10,5,3,2 K2048", 2048, [10,5,3,2], std),
                        ("all 10,5,3,2", 0, [10,5,3,2], [10,5,3,2]), ("all 8,5,3,2,1,1", 0, std, [8,5,3,2,1,1]), ("all 10,5,2,1,1,1", 0, std, [10,5,2,1,1,1]),
                        ("all 12,5,2,1", 0, std, [12,5,2,1])]:
    ms = [sim(c[k], plan_tickets(np.argsort(-c[k-1]), K, pA, pB)) for k in range(3, 12)]
    print(f"{name:22s} {np.mean(ms):.1f}")
