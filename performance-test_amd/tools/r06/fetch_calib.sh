#!/bin/bash
# FETCH_SIZE / WRITE_SIZE per known byte on this box (tools/micro/fetch_calib.hip).  GPU box: bash tools/r06/fetch_calib.sh
R=$GRAFT_REPO_ROOT
B=$R/performance-test_amd/tools/micro/fetch_calib
O=$R/gpurun_out/fetch_calib
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
$B > $O/plain.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -o p -- $B > $O/pmc_$c.log 2>&1
done
python3 - $O <<'PY'
import csv, glob, sys, collections, re
o = sys.argv[1]
line = open(o + "/plain.log").read()
moved = dict(re.findall(r"(k_\w+) (\d+)", line))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"{o}/pmc_{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = re.search(r"(k_\w+)", r["Kernel_Name"])
            if k and r["Counter_Name"] == c:
                acc[k.group(1)][c].append(float(r["Counter_Value"]))
out = ["kernel            bytes moved   FETCH_SIZE (KiB)  bytes / (FETCH KiB x 1024)   WRITE_SIZE (KiB)  bytes / (WRITE KiB x 1024)"]
for k in ("k_stream16", "k_stream16nt", "k_stream8", "k_pair16u", "k_rec24", "k_write16"):
    b = float(moved.get(k, 0))
    f = sorted(acc[k]["FETCH_SIZE"])[len(acc[k]["FETCH_SIZE"]) // 2] if acc[k]["FETCH_SIZE"] else 0.0
    w = sorted(acc[k]["WRITE_SIZE"])[len(acc[k]["WRITE_SIZE"]) // 2] if acc[k]["WRITE_SIZE"] else 0.0
    out.append(f"{k:16s} {b:13.0f} {f:16.1f} {(b / (f * 1024) if f else 0):26.3f} {w:18.1f} {(b / (w * 1024) if w else 0):26.3f}")
open(o + "/fetch_calibration.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out))
PY
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
