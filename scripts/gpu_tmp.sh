cd $GRAFT_REPO_ROOT
B="python bench.py --no-cpu-baseline --no-live-counters"
val() { grep '^{"metric"' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,3), d['valid'])"; }
for d in 2.0 1.8 1.667 1.5; do for rep in 1 2; do echo -n "div $d 20-step: "; NMF_CHUNK_DIV=$d timeout 200 $B --steps 20 --warmup 5 2>/dev/null | val; done
echo -n "div $d 50-step: "; NMF_CHUNK_DIV=$d timeout 200 $B 2>/dev/null | val; echo -n "div $d replay 20-step: "; NMF_CHUNK_DIV=$d timeout 200 $B --workload replay --steps 20 --warmup 5 2>/dev/null | val; done
