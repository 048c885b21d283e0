"""The oracle against its own frozen vectors (tests/golden/oracle_regression.npz, tests/golden/make_oracle_regression.py):
a guard on the oracle, the model compiler and the asset pack — not a claim about MuJoCo."""
import importlib.util
from pathlib import Path

import numpy as np

GOLD = Path(__file__).parent / "golden"


def test_oracle_reproduces_its_frozen_trajectories(oracle_lib):
    spec = importlib.util.spec_from_file_location("make_oracle_regression", GOLD / "make_oracle_regression.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    now = mod.trajectories()
    gold = np.load(GOLD / "oracle_regression.npz")
    assert set(gold.files) == set(now)
    for k in gold.files:
        if k.endswith("ncon_iters"):
            np.testing.assert_array_equal(now[k], gold[k])
        else:
            # float64 arithmetic, same compiler flags: differences only from libm / instruction selection on another host
            np.testing.assert_allclose(now[k], gold[k], rtol=0, atol=1e-7 * max(1.0, np.abs(gold[k]).max()), err_msg=k)
