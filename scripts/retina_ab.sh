cd $GRAFT_REPO_ROOT
python -m pytest tests/test_sensors.py -q -m gpu 2>&1 | tail -3
python scripts/bench_vision.py --steps 200 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel_ms'], d['roofline']['achieved'], d['value'])"
