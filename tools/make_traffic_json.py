"""Turns a tools/pmc_run.sh output directory into profiles/pmc_traffic.json: HBM bytes per FRAME (or training
step) per operator, from the FETCH_SIZE / WRITE_SIZE passes (KB per dispatch), corrected as
/opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950 (FETCH_SIZE x 2 for wide coalesced reads;
WRITE_SIZE as is).  bench.py reads this file for `roofline.traffic` and REFUSES an entry whose recorded kernel
symbols (`_kernels`) are not the ones it launches: the file goes stale silently when a kernel changes.

    python tools/make_traffic_json.py gpurun_out/pmc/fwd   profiles/pmc_traffic.json n1000000 FRAMES
    python tools/make_traffic_json.py gpurun_out/pmc/train profiles/pmc_traffic.json train_n1000000_1600x1066 STEPS train
"""
import csv, glob, json, os, re, sys
from collections import defaultdict

OPS_FWD = {
    "projection": ["projection_fwd_kernel"],
    "isect_tiles": ["bin_count_kernel", "center_scatter_kernel",
                    "bin_scatter_flat_kernel", "big_split_kernel", "super_sort_kernel"],
    "spherical_harmonics": ["sh_fwd_kernel"],
    "rasterize_to_pixels": ["raster_fwd_wave_kernel", "raster_fwd_ref_kernel"],
}
OPS_TRAIN = dict(OPS_FWD, **{
    "rasterize_to_pixels_bwd": ["raster_bwd_wave_kernel", "raster_bwd_kernel"],
    "spherical_harmonics_bwd": ["sh_bwd_kernel"],
    "projection_bwd": ["projection_bwd_kernel"],
})
src, dst, key, frames = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
OPS = OPS_TRAIN if len(sys.argv) > 5 and sys.argv[5] == "train" else OPS_FWD


def clean(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*$", "", name)          # keeps the template arguments: raster_fwd_wave_kernel<4, false, false, false>


tot = defaultdict(lambda: defaultdict(float))      # kernel base name -> counter -> sum over all dispatches (KB)
symbols = defaultdict(set)                          # kernel base name -> full symbols seen
for f in glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            c = r.get("Counter_Name", "")
            if c not in ("FETCH_SIZE", "WRITE_SIZE"):
                continue
            full = clean(r.get("Kernel_Name", ""))
            k = re.sub(r"<.*$", "", full)
            tot[k][c] += float(r["Counter_Value"])
            symbols[k].add(full)
out = json.load(open(dst)) if os.path.exists(dst) else {}
detail, kern = {}, {}
for op, kernels in OPS.items():
    b = 0.0
    for k in kernels:
        if k in tot:
            kb = 2.0 * tot[k]["FETCH_SIZE"] + tot[k]["WRITE_SIZE"]
            b += kb * 1024.0 / frames
            detail[k] = {"fetch_KB_per_frame": tot[k]["FETCH_SIZE"] / frames, "write_KB_per_frame": tot[k]["WRITE_SIZE"] / frames}
            kern.setdefault(op, []).extend(sorted(symbols[k]))
    out.setdefault(op, {})[key] = b
out.setdefault("_detail", {})[key] = detail
out.setdefault("_kernels", {})[key] = kern
out["_note"] = ("bytes per frame (per training step for train_* keys) = (2*FETCH_SIZE + WRITE_SIZE) KB * 1024 summed over the "
                "operator's kernels (gfx950: FETCH_SIZE under-reports wide coalesced reads by 2x, MI355X_MICROARCH.md HBM "
                "section); _kernels: the full kernel symbols the counters were collected on")
json.dump(out, open(dst, "w"), indent=1, sort_keys=True)
print(json.dumps({k: v for k, v in out.items() if not k.startswith("_")}, indent=1))
