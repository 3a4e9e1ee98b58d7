#!/usr/bin/env python3
"""Fold gpurun_out/parity.jsonl (written by the -m gpu tests through tests/parity_log.py) into profiles/rNN/parity.json:
the latest record per (test, quantity), sorted.  Usage: tools/collect_parity.py [round, default 02]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1] if len(sys.argv) > 1 else "02"
src = os.path.join(ROOT, "gpurun_out", "parity.jsonl")
last = {}
for line in open(src):
    line = line.strip()
    if line:
        r = json.loads(line)
        last[(r["test"], r["quantity"])] = r
out = [last[k] for k in sorted(last)]
dst = os.path.join(ROOT, "profiles", f"r{rnd}", "parity.json")
os.makedirs(os.path.dirname(dst), exist_ok=True)
json.dump({"rule": "e_hip <= max(tol, factor * e_ref); e_* are max-abs errors relative to the max-abs of the float64 result",
           "records": out}, open(dst, "w"), indent=1)
print(f"{len(out)} records -> {dst}")
worst = sorted(out, key=lambda r: -(r.get("e_hip") or 0) / max(r.get("bound") or 1e-30, 1e-30))[:10]
for r in worst:
    f = lambda v: "   n/a  " if v is None else f"{v:.2e}"
    print(f"  {r['test']:60s} {r['quantity']:24s} e_hip {f(r.get('e_hip'))}  e_ref {f(r.get('e_ref'))}  bound {f(r.get('bound'))}")
