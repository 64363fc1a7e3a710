"""Prints the ordered kernel sequence (start offset, duration, gap to the previous kernel) of the
last complete frame found in a rocprofv3 --kernel-trace CSV directory.

    python tools/trace_sequence.py gpurun_out/prof_dir [anchor_kernel_substring]
"""
import csv
import glob
import os
import re
import sys

src = sys.argv[1]
anchor = sys.argv[2] if len(sys.argv) > 2 else "projection_fwd_kernel"
rows = []
for f in glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
idx = [i for i, r in enumerate(rows) if anchor in r[2]]
if len(idx) < 3:
    sys.exit("not enough frames")
a, b = idx[-2], idx[-1]
t0 = rows[a][0]
prev_end = rows[a - 1][1] if a else t0
busy = 0
for s, e, n in rows[a:b]:
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)[:70]
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:7.1f}  gap {(s - prev_end) / 1e3:6.1f}  {n}")
    busy += e - s
    prev_end = e
print(f"frame span {(rows[b][0] - t0) / 1e3:.1f} us, busy {busy / 1e3:.1f} us")
