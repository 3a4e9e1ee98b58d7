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


def _strip_comments(text):
    """C / C++ source without comments and with runs of blanks collapsed (string and character literals kept verbatim)."""
    out, i, n = [], 0, len(text)
    while i < n:
        c = text[i]
        if c == '"' or c == "'":
            j = i + 1
            while j < n and text[j] != c:
                j += 2 if text[j] == "\\" else 1
            out.append(text[i:j + 1])
            i = j + 1
        elif text.startswith("//", i):
            j = text.find("\n", i)
            i = n if j < 0 else j
        elif text.startswith("/*", i):
            j = text.find("*/", i + 2)
            i = n if j < 0 else j + 2
            out.append(" ")
        else:
            out.append(c)
            i += 1
    return re.sub(r"[ \t]+", " ", re.sub(r"[ \t]*\n\s*", "\n", "".join(out))).strip()


def kernel_source_digest(root=ROOT):
    """sha256 over the kernel sources (nefes_amd/csrc: *.hip, *.h, *.cpp, Makefile) WITHOUT their comments, 16 hex digits.  Stored with
    every counter digest; bench.py quotes `roofline.traffic` from a committed digest only while this still matches the tree it runs
    from.  (Comments are stripped since round 6: a reworded comment is not a different kernel, and every such edit used to cost a
    re-collection; the Makefile's flags and every token of code still count.)"""
    d = os.path.join(root, "nefes_amd", "csrc")
    h = hashlib.sha256()
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h", ".cpp")) or f == "Makefile":
            h.update(f.encode())
            text = open(os.path.join(d, f), "r", errors="replace").read()
            h.update((text if f == "Makefile" else _strip_comments(text)).encode())
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
