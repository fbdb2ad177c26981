#!/usr/bin/env python3
"""usage: tools/merge_traffic.py <round dir, e.g. profiles/r02> <tag>=<key> ...
Copies gpurun_out/pmc_record_<tag>/{summary.txt,kernel_stats.csv,record.json} to
<round dir>/<tag>_{pmc_summary.txt,kernel_stats.csv,pmc_record.json} and merges the record into
profiles/traffic.json under <key> (bench.py looks up "<N>x<E>" or "<N>x<E>+<nb>b")."""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1]
os.makedirs(os.path.join(ROOT, rnd), exist_ok=True)
tpath = os.path.join(ROOT, "profiles", "traffic.json")
tr = json.load(open(tpath)) if os.path.exists(tpath) else {}
for spec in sys.argv[2:]:
    tag, key = spec.split("=")
    src = os.path.join(ROOT, "gpurun_out", "pmc_record_" + tag)
    for a, b in (("summary.txt", "pmc_summary.txt"), ("kernel_stats.csv", "kernel_stats.csv"), ("record.json", "pmc_record.json")):
        if os.path.exists(os.path.join(src, a)):
            shutil.copy(os.path.join(src, a), os.path.join(ROOT, rnd, f"{tag}_{b}"))
    rec = json.load(open(os.path.join(src, "record.json")))
    rec["source"] = f"{rnd}/{tag}_pmc_summary.txt"
    tr[key] = rec
    print(key, "hbm MB", rec["hbm_bytes"] / 1e6, "fp64 flop", rec["fp64_flops"])
json.dump(tr, open(tpath, "w"), indent=1)
