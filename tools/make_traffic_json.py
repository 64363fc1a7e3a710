"""Turns a tools/pmc_run.sh output directory into profiles/pmc_traffic.json: HBM bytes per FRAME per
operator, from the FETCH_SIZE / WRITE_SIZE passes (KB per dispatch), corrected as
/opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950 (FETCH_SIZE x 2 for wide coalesced
reads; WRITE_SIZE as is).  bench.py reads this file for `roofline.traffic`.

    python tools/make_traffic_json.py gpurun_out/pmc profiles/pmc_traffic.json n1000000 FRAMES
"""
import csv, glob, json, os, re, sys
from collections import defaultdict

OPS = {
    "projection": ["projection_fwd_kernel"],
    "isect_tiles": ["bin_count_kernel", "center_scatter_kernel",
                    "bin_scatter_flat_kernel", "big_split_kernel", "super_sort_kernel"],
    "spherical_harmonics": ["sh_fwd_kernel"],
    "rasterize_to_pixels": ["raster_fwd_wave_kernel", "raster_fwd_ref_kernel"],
}
src, dst, key, frames = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
tot = defaultdict(lambda: defaultdict(float))      # kernel -> counter -> sum over all dispatches (KB)
for f in glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            c = r.get("Counter_Name", "")
            if c not in ("FETCH_SIZE", "WRITE_SIZE"):
                continue
            k = re.sub(r"\(anonymous namespace\)::", "", r.get("Kernel_Name", ""))
            k = re.sub(r"^void ", "", k)
            k = re.sub(r"[<(].*$", "", k)
            tot[k][c] += float(r["Counter_Value"])
out = json.load(open(dst)) if os.path.exists(dst) else {}
detail = {}
for op, kernels in OPS.items():
    b = 0.0
    for k in kernels:
        if k in tot:
            kb = 2.0 * tot[k]["FETCH_SIZE"] + tot[k]["WRITE_SIZE"]
            b += kb * 1024.0 / frames
            detail[k] = {"fetch_KB_per_frame": tot[k]["FETCH_SIZE"] / frames, "write_KB_per_frame": tot[k]["WRITE_SIZE"] / frames}
    out.setdefault(op, {})[key] = b
out.setdefault("_detail", {})[key] = detail
out["_note"] = ("bytes per frame = (2*FETCH_SIZE + WRITE_SIZE) KB * 1024 summed over the operator's kernels "
                "(gfx950: FETCH_SIZE under-reports wide coalesced reads by 2x, MI355X_MICROARCH.md HBM section)")
json.dump(out, open(dst, "w"), indent=1, sort_keys=True)
print(json.dumps({k: v for k, v in out.items() if not k.startswith("_")}, indent=1))
