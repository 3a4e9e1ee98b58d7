"""Persist the parity numbers the GPU tests measure: every three-way comparison

    e_hip  = |hip  - float64 oracle| / scale
    e_ref  = |fp32 reference/oracle - float64 oracle| / scale      (the reference's own fp32 noise floor)
    direct = |hip  - fp32 reference| / scale

is appended as one JSON line to $NEFES_PARITY_LOG (default gpurun_out/parity.jsonl; gpurun merges that directory back).
tools/collect_parity.py folds the lines into profiles/rNN/parity.json, which is tracked.  The acceptance rule the tests
share lives here too, so that every test states its bound the same way."""
import json
import os
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATH = os.environ.get("NEFES_PARITY_LOG", os.path.join(ROOT, "gpurun_out", "parity.jsonl"))

NORTH_STAR_TOL = 1e-4          # BASELINE.json: maps and pose gradients within 1e-4 relative (fp32)
REF_FACTOR = 1.5               # ... or no further from the float64 truth than 1.5 x the reference's own fp32 arithmetic


def record(test, quantity, **numbers):
    rec = {"test": test, "quantity": quantity, "time": time.strftime("%Y-%m-%dT%H:%M:%S")}
    rec.update({k: (float(v) if v is not None else None) for k, v in numbers.items()})
    try:
        os.makedirs(os.path.dirname(PATH), exist_ok=True)
        with open(PATH, "a") as f:
            f.write(json.dumps(rec) + "\n")
    except OSError:
        pass
    return rec


def bound(e_ref, tol=NORTH_STAR_TOL, factor=REF_FACTOR):
    """Largest accepted distance from the float64 truth: the north-star tolerance, or `factor` x the distance of the
    reference's own fp32 result from that truth where fp32 arithmetic itself is further away than the tolerance."""
    return max(tol, factor * e_ref)


def check(test, quantity, e_hip, e_ref, direct=None, tol=NORTH_STAR_TOL, factor=REF_FACTOR):
    record(test, quantity, e_hip=e_hip, e_ref=e_ref, direct=direct, bound=bound(e_ref, tol, factor))
    assert e_hip <= bound(e_ref, tol, factor), (test, quantity, e_hip, e_ref)
