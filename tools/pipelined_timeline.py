#!/usr/bin/env python3
"""Timeline of a rocprofv3 kernel trace of tools/pipelined_probe.py: the amdmsm kernels of the last few MSMs with start /
end relative to the first listed kernel, the queue they ran on, and for every tail kernel (fix-up, reduction, Horner) the
share of its duration during which a k_accumulate or a sort kernel of ANOTHER MSM was running.

  python tools/pipelined_timeline.py <kernel_trace.csv> [msms_to_show]
"""
import csv
import re
import sys

BULK = ("k_accumulate", "k_sort_digits", "k_sort_scan", "k_sort_coarse", "k_sort_fine", "k_sort_big_hist", "k_sort_big_scan",
        "k_sort_big_scatter", "k_endo_points")
TAIL = ("k_accumulate_fixup", "k_accumulate_compact", "k_accumulate_fixup_queue", "k_bucket_sums", "k_plane_sums",
        "k_window_horner", "k_horner", "k_reduce_segments", "k_sum_butterfly", "k_sum_block", "k_sum_block_wide")


def main():
    rows = []
    for r in csv.DictReader(open(sys.argv[1])):
        m = re.search(r"amdmsm::.*?::(k_\w+)", r["Kernel_Name"])
        if not m:
            continue
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), m.group(1), r.get("Queue_Id", "?")))
    rows.sort()
    show = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    # an MSM ends with k_horner
    ends = [i for i, r in enumerate(rows) if r[2] == "k_horner"]
    first = ends[-show - 1] + 1 if len(ends) > show else 0
    # start from the first sort kernel after that
    sel = rows[first:]
    t0 = sel[0][0]
    acc = [(s, e) for s, e, k, q in rows if k == "k_accumulate"]
    bulk = [(s, e) for s, e, k, q in rows if k in BULK]

    def overlap(s, e, ivs):
        tot = 0
        for a, b in ivs:
            lo, hi = max(s, a), min(e, b)
            if hi > lo:
                tot += hi - lo
        return tot

    print(f"{'kernel':28s} {'queue':>6s} {'start us':>10s} {'end us':>10s} {'dur us':>8s}  beside k_accumulate / any bulk kernel")
    for s, e, k, q in sel:
        extra = ""
        if k in TAIL:
            d = max(e - s, 1)
            extra = f"{100.0 * overlap(s, e, acc) / d:5.0f} % / {100.0 * min(overlap(s, e, bulk), d) / d:5.0f} %"
        print(f"{k:28s} {q:>6s} {(s - t0) / 1e3:10.1f} {(e - t0) / 1e3:10.1f} {(e - s) / 1e3:8.1f}  {extra}")
    hs = [r[1] for r in rows if r[2] == "k_horner"]
    if len(hs) > 4:
        per = (hs[-1] - hs[-5]) / 4 / 1e3
        print(f"\nk_horner to k_horner, mean of the last four MSMs: {per:.1f} us per MSM")


if __name__ == "__main__":
    main()
