#!/usr/bin/env python3
"""Turn the rocprofv3 outputs a gpurun call left under gpurun_out/ into the tracked summaries
under profiles/ (kernel-trace stats of the amdmsm kernels; FETCH_SIZE / WRITE_SIZE per kernel).

  python tools/summarize_profiles.py <tag> <kernel_stats.csv> <fetch_counter_collection.csv> \
         <write_counter_collection.csv> <curve> <group> <log2n> <window_bits>
"""
import collections
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    tag, stats, fetch, write, curve, group, log2n, c = sys.argv[1:9]
    rows = list(csv.reader(open(stats)))
    out = [rows[0]] + [r for r in rows[1:] if "amdmsm" in r[0]]
    csv.writer(open(os.path.join(ROOT, "profiles", f"{tag}_kernel_stats.csv"), "w")).writerows(out)
    res = {}
    for name, f in (("FETCH_SIZE", fetch), ("WRITE_SIZE", write)):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            m = re.search(r"::(k_\w+)", r["Kernel_Name"])
            if m and "amdmsm" in r["Kernel_Name"]:
                agg[m.group(1)].append(float(r["Counter_Value"]))
        for k, v in sorted(agg.items()):
            res.setdefault(k, {})[name] = sum(v) / len(v)
    doc = {
        "workload": {"curve": curve, "group": int(group), "log2n": int(log2n), "window_bits": int(c)},
        "command": "rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -- python3 bench.py --steps K --warmup 1 "
                   f"--no-cpu-baseline --no-legs --log2n {log2n}   (second pass: --pmc WRITE_SIZE); tools/collect_profiles.sh",
        "units": "FETCH_SIZE / WRITE_SIZE as reported by rocprofv3 (KB per dispatch, mean over the dispatches of the run)",
        "correction": "MI355X_MICROARCH.md HBM section: on gfx950 FETCH_SIZE counts 64 B per 128-B request, i.e. half the "
                      "bytes of 16-B-per-lane reads -> traffic = (2*FETCH_SIZE + WRITE_SIZE) * 1024; Infinity-Cache hits "
                      "are included in FETCH_SIZE",
        "kernels": res,
    }
    json.dump(doc, open(os.path.join(ROOT, "profiles", f"{tag}_pmc.json"), "w"), indent=1)
    k = res.get("k_accumulate", {})
    print("k_accumulate traffic bytes:", (2 * k.get("FETCH_SIZE", 0) + k.get("WRITE_SIZE", 0)) * 1024)


if __name__ == "__main__":
    main()
