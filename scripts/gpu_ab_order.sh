cd $GRAFT_REPO_ROOT
B="python bench.py --no-cpu-baseline --no-live-counters"
for a in "--steps-per-launch 250" "--steps 20 --warmup 5" "" "--steps-per-launch 100"; do echo "== $a"; timeout 300 $B $a 2>/dev/null | grep '^{"metric"' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,3))"; done
