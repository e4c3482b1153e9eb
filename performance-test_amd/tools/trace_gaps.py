#!/usr/bin/env python3
"""Timeline of the CG loop from a rocprofv3 --kernel-trace CSV: per kernel of the iteration its mean
duration and the mean idle gap before it, over the dispatches of the solve's steady state.

  trace_gaps.py <dir with *kernel_trace.csv> [out.csv]

The loop is recognised by its SpMV dispatches (spmv_sellp_kernel / spmv_tile_kernel); dispatches between two
consecutive SpMV launches form one iteration.  Iterations after convergence (SpMV shorter than 3 us) are dropped."""
import csv
import glob
import os
import re
import sys
from collections import OrderedDict


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    name = re.sub(r"<.*$", "", name)
    return name.replace("zzz::", "")


def main():
    d = sys.argv[1]
    files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    rows = []
    for f in files:
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
    rows.sort()
    is_spmv = lambda n: n.startswith("spmv_")  # noqa: E731
    idx = [i for i, r in enumerate(rows) if is_spmv(r[2]) and r[1] - r[0] > 3000]
    # an iteration = dispatches from one working SpMV (exclusive) to the next (inclusive)
    acc = OrderedDict()
    iters = []
    for a, b in zip(idx[:-1], idx[1:]):
        seq = rows[a + 1:b + 1]
        names = tuple(r[2] for r in seq)
        if len(seq) > 12:  # a solve boundary (assembly, pattern ...) lies in between
            continue
        iters.append((names, rows[b][1] - rows[a][1], seq, rows[a][1]))
    if not iters:
        print("no CG iterations found")
        return
    # the most frequent iteration shape
    from collections import Counter
    shape = Counter(i[0] for i in iters).most_common(1)[0][0]
    sel = [i for i in iters if i[0] == shape]
    n = len(sel)
    period = sum(i[1] for i in sel) / n
    print(f"iterations of the dominant shape: {n} of {len(iters)}; period {period / 1e3:.2f} us")
    out = [("kernel", "mean_duration_us", "mean_gap_before_us")]
    tot_d = tot_g = 0.0
    for k, name in enumerate(shape):
        dur = sum(i[2][k][1] - i[2][k][0] for i in sel) / n
        gap = sum(i[2][k][0] - (i[2][k - 1][1] if k else i[3]) for i in sel) / n
        tot_d += dur
        tot_g += gap
        out.append((name, f"{dur / 1e3:.2f}", f"{gap / 1e3:.2f}"))
        print(f"  {name:28s} {dur / 1e3:8.2f} us   gap before {gap / 1e3:6.2f} us")
    out.append(("TOTAL", f"{tot_d / 1e3:.2f}", f"{tot_g / 1e3:.2f}"))
    out.append(("period", f"{period / 1e3:.2f}", ""))
    print(f"  kernels {tot_d / 1e3:.2f} us + gaps {tot_g / 1e3:.2f} us")
    if len(sys.argv) > 2:
        with open(sys.argv[2], "w", newline="") as fh:
            csv.writer(fh).writerows(out)


if __name__ == "__main__":
    main()
