#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc output (one *_counter_collection.csv per process) per kernel and counter.

  python3 pmc_summary.py DIR [DIR ...] [--match SUBSTR] [--traffic-json OUT --key KEY --kernel SUBSTR --source TEXT]

Prints `kernel  counter  launches  mean  min  max` (counters summed over the dispatch's rows, i.e. over XCDs /
instances as rocprofv3 reports them).  With --traffic-json it also writes the HBM traffic record bench.py
reads: read = 2 x FETCH_SIZE KiB (gfx950 correction, MI355X_MICROARCH.md "HBM"), write = WRITE_SIZE KiB, with
the sha256 of the kernel sources so that bench.py can refuse a stale record."""
import argparse
import collections
import csv
import glob
import hashlib
import json
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def short(name):
    name = re.sub(r"^void ", "", name).replace("(anonymous namespace)::", "")
    return re.sub(r"\(.*$", "", name)


def kernel_source_hash():
    h = hashlib.sha256()
    # the sources that define the timed batch kernel and its tables (same list as bench.py)
    for name in ("zk_sep_patches.hip", "zk_sep.h", "zk_sep.hip", "zk_fold.h", "zk_internal.h"):
        h.update(open(os.path.join(ROOT, "motif-learn_amd", "csrc", name), "rb").read())
    return h.hexdigest()[:16]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dirs", nargs="+")
    ap.add_argument("--match", default="")
    ap.add_argument("--traffic-json")
    ap.add_argument("--key")
    ap.add_argument("--kernel")
    ap.add_argument("--source", default="")
    args = ap.parse_args()
    per = collections.defaultdict(lambda: collections.defaultdict(float))      # (kernel, counter) -> dispatch -> sum
    for d in args.dirs:
        for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(path) as f:
                for row in csv.DictReader(f):
                    k = short(row["Kernel_Name"])
                    if args.match in k:
                        per[(k, row["Counter_Name"])][(path, row["Dispatch_Id"])] += float(row["Counter_Value"])
    stats = {}
    for (k, c), disp in sorted(per.items()):
        v = list(disp.values())
        stats[(k, c)] = (len(v), sum(v) / len(v), min(v), max(v))
        print(f"{k[:70]:70s} {c:22s} launches={len(v):4d} mean={stats[(k, c)][1]:16.1f} min={min(v):16.1f} max={max(v):16.1f}")
    if args.traffic_json:
        pick = lambda c: [s for (k, cc), s in stats.items() if cc == c and args.kernel in k]
        fetch, write = pick("FETCH_SIZE"), pick("WRITE_SIZE")
        if not fetch or not write:
            raise SystemExit("need FETCH_SIZE and WRITE_SIZE rows for the kernel")
        rec = {"read_bytes": 2.0 * fetch[0][1] * 1024, "write_bytes": write[0][1] * 1024}
        rec["hbm_bytes_per_launch"] = rec["read_bytes"] + rec["write_bytes"]
        rec["source"] = args.source
        rec["kernel_source_sha"] = kernel_source_hash()
        try:
            rec["git"] = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], text=True).strip()
        except Exception:
            rec["git"] = None
        data = {}
        if os.path.exists(args.traffic_json):
            data = json.load(open(args.traffic_json))
        data[args.key] = rec
        json.dump(data, open(args.traffic_json, "w"), indent=1)
        print("wrote", args.traffic_json, args.key, rec)


if __name__ == "__main__":
    main()
