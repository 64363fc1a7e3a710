"""How well do two frames in flight overlap?  From a rocprofv3 --kernel-trace CSV directory of the headline loop:
wall time per frame against the sum of kernel durations, the share of the wall time with 0 / 1 / 2+ kernels running,
and per kernel its mean duration (to set beside the one-frame-in-flight profile: a kernel that shares the GPU runs longer).

    python tools/trace_overlap.py gpurun_out/prof_dir [frames_to_skip_at_both_ends]
"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict

src = sys.argv[1]
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 8
rows = []
for f in glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
anchors = [i for i, r in enumerate(rows) if "projection_fwd_kernel" in r[2]]
if len(anchors) < 2 * skip + 4:
    sys.exit("not enough frames")
a, b = anchors[skip], anchors[-skip]
frames = len(anchors) - 2 * skip
win = rows[a:b]
t0, t1 = win[0][0], rows[b][0]
ev = []
for s, e, _ in win:
    ev.append((s, 1)); ev.append((min(e, t1), -1))
ev.sort()
depth, last, hist = 0, t0, defaultdict(int)
for t, d in ev:
    hist[min(depth, 3)] += t - last
    last = t
    depth += d
wall = (t1 - t0) / 1e3
ksum = sum(e - s for s, e, _ in win) / 1e3
print(f"{frames} frames: wall {wall / frames:.1f} us per frame, sum of kernel durations {ksum / frames:.1f} us per frame "
      f"(ratio {ksum / wall:.2f})")
print("share of the wall time with k kernels running: " +
      ", ".join(f"{k}{'+' if k == 3 else ''}: {100 * hist[k] / 1e3 / wall:.1f} %" for k in range(4)))
per = defaultdict(list)
for s, e, n in win:
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    per[n.split("(")[0][:80]].append((e - s) / 1e3)
print("| kernel | launches / frame | mean us | us / frame |\n|---|---|---|---|")
for n, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
    if sum(v) / frames < 1.0:
        continue
    print(f"| `{n}` | {len(v) / frames:.2f} | {sum(v) / len(v):.1f} | {sum(v) / frames:.1f} |")
