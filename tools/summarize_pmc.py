"""Averages rocprofv3 --pmc counter CSVs per kernel (value per launch) -> markdown table."""
import csv, glob, os, re, sys
from collections import defaultdict

def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*$", "", name)[:60]

src, dst = sys.argv[1], sys.argv[2]
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for f in glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            k = short(r.get("Kernel_Name", ""))
            c = r.get("Counter_Name", "")
            try:
                v = float(r.get("Counter_Value", "nan"))
            except ValueError:
                continue
            # one row per (dispatch, counter[, dimension]); sum dimensions of a dispatch, average dispatches
            key = (r.get("Dispatch_Id", ""), c)
            acc[k][c][0] += v
            acc[k][c].append(key) if False else None
            acc[k][c][1] += 0
disp = defaultdict(lambda: defaultdict(set))
for f in glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            disp[short(r.get("Kernel_Name", ""))][r.get("Counter_Name", "")].add(r.get("Dispatch_Id", ""))
counters = sorted({c for k in acc for c in acc[k]})
lines = ["| kernel | " + " | ".join(counters) + " |", "|---|" + "---|" * len(counters)]
for k in sorted(acc, key=lambda k: -acc[k].get("SQ_WAVE_CYCLES", [0])[0]):
    row = []
    for c in counters:
        n = max(len(disp[k][c]), 1)
        row.append(f"{acc[k][c][0] / n:.4g}" if c in acc[k] else "")
    lines.append(f"| `{k}` | " + " | ".join(row) + " |")
open(dst, "w").write("\n".join(lines) + "\n")
print("\n".join(lines[:14]))
