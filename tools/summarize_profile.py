#!/usr/bin/env python3
"""Turn gpurun_out/prof_<tag>/ (tools/profile_bench.sh) into the tracked summaries under profiles/:
   <tag>_kernel_stats.csv  rocprofv3 --kernel-trace --stats table
   <tag>_pmc.json          per-kernel mean counters + derived HBM traffic
   pmc_traffic.json        what bench.py reports as roofline.traffic for the same workload
HBM bytes follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE and WRITE_SIZE are in KiB;
on gfx950 FETCH_SIZE counts half the bytes of a coalesced stream -- verified for this kernel's own access
shape with tools/fetch_calib.hip (ratio 0.5000 for 8 B/lane windowed buffer loads and for 16 B/lane streams),
so bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024."""
import collections, csv, glob, json, os, shutil, sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
src = os.path.join(root, "gpurun_out", "prof_" + tag)
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)
shutil.copy(glob.glob(os.path.join(src, "stats", "*", "*_kernel_stats.csv"))[0], os.path.join(dst, tag + "_kernel_stats.csv"))
out = collections.defaultdict(dict)
for sub in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_sq2"):
    fs = glob.glob(os.path.join(src, sub, "*", "*_counter_collection.csv"))
    if not fs:
        continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        agg[(r["Kernel_Name"].split("(")[0][:70], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in agg.items():
        out[k][c] = sum(v) / len(v)
        out[k]["launches_" + sub] = len(v)
bench = {}
bl = os.path.join(src, "bench_line.json")
if os.path.exists(bl) and os.path.getsize(bl):
    bench = json.loads(open(bl).read())
for k, c in out.items():
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        c["hbm_bytes_per_launch"] = (2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024
json.dump({"bench_line": bench, "kernels": out}, open(os.path.join(dst, tag + "_pmc.json"), "w"), indent=1, sort_keys=True)
spec = [c for k, c in out.items() if "spectrum_kernel" in k]
if spec and bench:
    cfg = bench["config"]
    rec = {"frames": cfg["frames_per_gpu_per_step"], "fmt": "c64" if cfg["input"] == "complex64" else "u8",
           "hbm_bytes_per_launch": spec[0]["hbm_bytes_per_launch"], "source": "profiles/%s_pmc.json" % tag,
           "fetch_size_kib": spec[0]["FETCH_SIZE"], "write_size_kib": spec[0]["WRITE_SIZE"]}
    json.dump(rec, open(os.path.join(dst, "pmc_traffic.json"), "w"), indent=1)
    alg = bench["roofline"]["algorithmic_bytes_per_launch"]
    print("traffic %.3f GB per launch vs algorithmic %.3f GB (x%.2f)" % (rec["hbm_bytes_per_launch"] / 1e9, alg / 1e9, rec["hbm_bytes_per_launch"] / alg))
print(open(os.path.join(dst, tag + "_kernel_stats.csv")).read()[:900])
