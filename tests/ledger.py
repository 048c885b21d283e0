"""Parity ledger (test helper; round-5 verdict item 1c: "no bar may loosen; report the tight per-state bars as counts so a
drift is visible in the log").

A test that compares populations, or allows a few states beyond a bar, calls :func:`report` with the figures it saw — among them
the counts against the *tight* bars of earlier rounds.  Every record is printed (``pytest -s`` / the captured log of a failing
test) and appended to ``gpurun_out/parity_ledger.jsonl`` when that directory is writable (the GPU box); the builder commits a copy
per round as ``profiles/rNN_parity_ledger.jsonl``, so two rounds' figures can be laid side by side.
"""

from __future__ import annotations

import json
import os
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]


def report(test: str, **figures) -> dict:
    rec = {"test": test}
    for k, v in figures.items():
        try:
            rec[k] = v.item() if hasattr(v, "item") else v
        except Exception:
            rec[k] = repr(v)
    line = json.dumps(rec, default=float)
    print("PARITY-LEDGER " + line)
    out = Path(os.environ.get("NMF_LEDGER_DIR", ROOT / "gpurun_out"))
    try:
        out.mkdir(exist_ok=True)
        with open(out / "parity_ledger.jsonl", "a") as f:
            f.write(line + "\n")
    except OSError:
        pass
    return rec
