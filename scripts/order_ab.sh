cd $GRAFT_REPO_ROOT
for wl in cpg replay; do
  python bench.py --no-cpu-baseline --no-live-counters --workload $wl 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$wl %.4e' % d['value'], d['roofline']['kernel_ms_per_launch'])"
done
python bench.py --no-cpu-baseline --no-live-counters --terrain blocks 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('blocks %.4e' % d['value'], d['roofline']['kernel_ms_per_launch'])"
python -m pytest tests -q -m gpu 2>&1 | tail -1
