"""Condenses a rocprofv3 output directory into a small per-kernel summary (committed under profiles/).

    python tools/summarize_rocprof.py gpurun_out/prof profiles/r01_kernel_stats.md [n_frames]

Reads *kernel_stats.csv (from `rocprofv3 --kernel-trace --stats --output-format csv`) or, failing
that, aggregates *kernel_trace.csv itself.  With n_frames the per-frame time of every kernel is
also printed (total / frames).
"""
from __future__ import annotations

import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name[:90]


def main():
    src, dst = sys.argv[1], sys.argv[2]
    frames = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    rows = defaultdict(lambda: [0, 0.0, 1e30, 0.0])   # calls, total_ns, min, max
    traces = glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True)
    for f in traces:
        with open(f) as fh:
            for r in csv.DictReader(fh):
                name = r.get("Kernel_Name") or r.get("kernel_name") or ""
                try:
                    dur = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
                except Exception:
                    continue
                a = rows[short(name)]
                a[0] += 1
                a[1] += dur
                a[2] = min(a[2], dur)
                a[3] = max(a[3], dur)
    if not rows:
        print("no kernel trace found under", src)
        sys.exit(1)
    total = sum(a[1] for a in rows.values())
    lines = ["| kernel | calls | total ms | avg us | min us | max us | % |" + (" us/frame |" if frames else ""),
             "|---|---|---|---|---|---|---|" + ("---|" if frames else "")]
    for name, a in sorted(rows.items(), key=lambda kv: -kv[1][1]):
        line = f"| `{name}` | {a[0]} | {a[1] / 1e6:.3f} | {a[1] / a[0] / 1e3:.2f} | {a[2] / 1e3:.2f} | " \
               f"{a[3] / 1e3:.2f} | {100 * a[1] / total:.1f} |"
        if frames:
            line += f" {a[1] / frames / 1e3:.1f} |"
        lines.append(line)
    lines.append("")
    lines.append(f"total kernel time {total / 1e6:.3f} ms" + (f" = {total / frames / 1e3:.1f} us/frame over {frames} frames" if frames else ""))
    os.makedirs(os.path.dirname(dst) or ".", exist_ok=True)
    with open(dst, "w") as fh:
        fh.write("\n".join(lines) + "\n")
    print("\n".join(lines[:40]))


if __name__ == "__main__":
    main()
