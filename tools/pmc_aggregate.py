#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc passes (counter_collection.csv files) into one per-kernel, per-launch JSON.

usage: pmc_aggregate.py out.json pass1_dir pass2_dir ...
Each pass directory is searched recursively for *counter_collection.csv.  Counter values are summed over the
dispatches of a kernel and divided by the number of dispatches, so every figure is "per launch" like roofline.achieved.
"""
import csv
import glob
import hashlib
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_source_digest(root=ROOT):
    """sha256 over the kernel sources (nefes_amd/csrc: *.hip, *.h, *.cpp, Makefile), 16 hex digits.  Stored with every counter
    digest; bench.py quotes `roofline.traffic` from a committed digest only while this still matches the tree it runs from."""
    d = os.path.join(root, "nefes_amd", "csrc")
    h = hashlib.sha256()
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h", ".cpp")) or f == "Makefile":
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def short(name):
    name = re.sub(r"^void ", "", name).replace("(anonymous namespace)::", "")
    return re.sub(r"\(.*$", "", name)


def main():
    out, dirs = sys.argv[1], sys.argv[2:]
    agg = {}
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            per_pass = {}
            for row in csv.DictReader(open(f)):
                k = short(row["Kernel_Name"])
                if "nefes" not in row["Kernel_Name"] and "_kernel" not in k:
                    continue
                e = per_pass.setdefault(k, {"ids": set(), "sums": {}})
                e["ids"].add(row["Dispatch_Id"])
                e["sums"][row["Counter_Name"]] = e["sums"].get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
            for k, e in per_pass.items():
                n = max(1, len(e["ids"]))
                a = agg.setdefault(k, {})
                a["_dispatches"] = n
                for c, v in e["sums"].items():
                    a[c] = v / n
    res = dict(sorted(agg.items()))
    res["_meta"] = {"kernel_sources": kernel_source_digest()}
    json.dump(res, open(out, "w"), indent=1)


if __name__ == "__main__":
    main()
