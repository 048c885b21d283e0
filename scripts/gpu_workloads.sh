#!/bin/bash
# the workload table of DESIGN.md §3: batch-size sweep under the reference protocol + the other configurations
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
B="python bench.py --no-cpu-baseline --no-live-counters --no-other-configs"
line() { grep '^{"metric"' | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); c = d['config']
    print(sys.argv[1], round(d['value'] / 1e6, 3), 'M', 'ms/launch', round(d['roofline']['kernel_ms_per_launch'], 3), 'contacts', round(c['mean_contacts'], 2), 'iters', round(c['mean_newton_iters'], 2), 'valid', d.get('valid'))
" "$1"; }
{
for n in 16 64 256 1024 2048 4096 8192 16384; do timeout 300 $B --workload replay --worlds-per-gpu $n 2>/dev/null | line "replay worlds $n"; done
# the reference benchmark's second series: every collision geom a capsule (run_gpu_benchmark.py:15-24 sweeps simplify_geom in [False, True])
for n in 16 64 256 1024 2048 4096 8192 16384; do timeout 300 $B --workload replay --simplify-geom --worlds-per-gpu $n 2>/dev/null | line "replay, all-capsule geoms, worlds $n"; done
for t in gapped blocks mixed; do timeout 300 $B --terrain $t 2>/dev/null | line "cpg terrain $t"; done
timeout 300 $B --terrain mixed --odor --cpg-adhesion 20 2>/dev/null | line "config5 mixed+odor+adhesion 4096"
timeout 300 $B --terrain mixed --odor --cpg-adhesion 20 --worlds-per-gpu 128 2>/dev/null | line "config5 mixed+odor+adhesion 128"
timeout 300 $B --joint-preset legs_active_only 2>/dev/null | line "legs_active_only"
timeout 300 $B --joint-preset all_biological 2>/dev/null | line "all_biological"
timeout 300 $B --joint-preset all_possible 2>/dev/null | line "all_possible"
timeout 300 $B --joint-preset all_biological --terrain mixed 2>/dev/null | line "all_biological, mixed terrain"
timeout 300 $B --terrain mixed --odor --cpg-adhesion 20 --worlds-per-gpu 1024 2>/dev/null | line "config5 mixed+odor+adhesion 1024"
timeout 300 $B --steps-per-launch 250 2>/dev/null | line "cpg 250-step launches"
timeout 300 $B --steps 20 --warmup 5 2>/dev/null | line "cpg 20-step launches (driver args)"
timeout 300 $B --workload replay --steps 20 --warmup 5 2>/dev/null | line "replay 20-step launches"
timeout 300 $B --vision resample --steps 200 2>/dev/null | line "config 3, vision resample"
timeout 300 $B --vision render --steps 200 2>/dev/null | line "config 3, vision render"
timeout 300 $B --vision render --eye-rays 16 --steps 200 2>/dev/null | line "config 3, vision render, 16 rays per ommatidium"
timeout 300 $B --obs-every 1 2>/dev/null | line "cpg, observation block recorded every step (50-step launches)"
timeout 300 $B --obs-every 1 --steps 20 --warmup 5 2>/dev/null | line "cpg, observation block recorded every step (20-step launches)"
timeout 300 $B --obs-every 10 2>/dev/null | line "cpg, observation block recorded every 10th step"
timeout 300 $B --steps-per-launch 1 --steps 100 2>/dev/null | line "cpg, one launch per step"
for n in 512 1024 1536 2048; do timeout 300 $B --worlds-per-gpu $n 2>/dev/null | line "cpg worlds $n"; done
timeout 300 $B --joint-preset all_possible --terrain mixed 2>/dev/null | line "all_possible, mixed terrain"
} > gpurun_out/workloads.log 2>&1
cat gpurun_out/workloads.log
