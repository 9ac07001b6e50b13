#!/usr/bin/env python3
"""Condense gpurun_out/prof_<tag> (written by tools/profile_bench.sh on the GPU box) into profiles/<tag>_*:
  <tag>_kernel_stats.csv  the rocprofv3 --kernel-trace --stats table
  <tag>_pmc.json          per-kernel mean counters + the bench line of the same run
and record the dominant kernels' HBM traffic per bench step in profiles/pmc_traffic.json under the key
"<config>:<fmt>:<units per step>", stamped with the hash of the kernel sources (bench.py reports it as
roofline.traffic only while that hash still matches).  HBM bytes follow MI355X_MICROARCH.md (HBM / rocprofv3):
FETCH_SIZE counts 64 B per 128-B request on gfx950 -> doubled; WRITE_SIZE is exact for 16 B/lane streams; both in KiB.
The x2 was re-checked for this code's own access shapes with tools/fetch_calib.hip (ratio 0.5000 for 8 B/lane
windowed buffer loads and for 16 B/lane streams), so bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024."""
import collections, csv, glob, json, os, shutil, sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
tag = sys.argv[1]
src = os.path.join(root, "gpurun_out", "prof_" + tag)
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)
newest = lambda pattern: max(glob.glob(pattern), key=os.path.getmtime)   # (gpurun merges every call's files into the same directories)
shutil.copy(newest(os.path.join(src, "stats", "*", "*_kernel_stats.csv")), os.path.join(dst, tag + "_kernel_stats.csv"))
out = collections.defaultdict(dict)
for sub in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_sq2"):
    fs = glob.glob(os.path.join(src, sub, "*", "*_counter_collection.csv"))
    if not fs:
        continue
    fs = [max(fs, key=os.path.getmtime)]
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
json.dump({"bench_line": bench, "bench_args": open(os.path.join(src, "bench_args.txt")).read().strip() if os.path.exists(os.path.join(src, "bench_args.txt")) else "",
           "kernels": out}, open(os.path.join(dst, tag + "_pmc.json"), "w"), indent=1, sort_keys=True)
# dominant kernels = the spectrum stage: spectrum_kernel (+ combine_parts) or the four-step kernels
dom = {k: c for k, c in out.items() if any(s in k for s in ("spectrum_kernel", "spectrum_pair_kernel", "spectrum32_kernel", "spectrum64_kernel", "combine_parts", "dif16_", "dif_wide_")) and "hbm_bytes_per_launch" in c}
if dom and bench:
    import bench as B
    cfg = bench["config"]
    units = cfg.get("frames_per_gpu_per_step") or cfg["passes_per_step"] * cfg["bands_on_rank0"]
    fmt = "c64" if cfg["input"] == "complex64" else "u8"
    steps_in_pmc_run = 4        # profile_bench.sh: --steps 3 --warmup 1
    total = sum(c["hbm_bytes_per_launch"] * c["launches_pmc_fetch"] for c in dom.values())
    per_step = total / steps_in_pmc_run
    path = os.path.join(dst, "pmc_traffic.json")
    rec = {"entries": {}}
    if os.path.exists(path):
        try:
            old = json.load(open(path))
            if "entries" in old:
                rec = old
        except Exception:
            pass
    key = "%d:%s:%d" % (cfg["baseline_config"], fmt, units)
    ent = {"hbm_bytes_per_launch": per_step, "csrc_sha256": B.csrc_sha256(), "source": "profiles/%s_pmc.json" % tag,
           "kernels": sorted(dom)}
    # what the stage keeps busy (MI355X_MICROARCH.md, LDS / counters): LDS-array cycles and VALU issue slots over the
    # stage's shader cycles (GRBM_GUI_ACTIVE counts every XCD: / 8), on all 256 CUs x 4 SIMDs (2 clk per wave64 fp32 issue)
    NUM_CU = 256
    sq = [c for c in dom.values() if "GRBM_GUI_ACTIVE" in c and "SQ_LDS_IDX_ACTIVE" in c and "SQ_INSTS_VALU" in c]
    if sq:
        cyc = sum(c["GRBM_GUI_ACTIVE"] / 8.0 * c["launches_pmc_sq"] for c in sq)
        lds = sum(c["SQ_LDS_IDX_ACTIVE"] * c["launches_pmc_sq2"] for c in sq)
        ent["lds_frac"] = lds / (NUM_CU * cyc)
        ent["lds_conflict_ratio"] = (sum(c["SQ_LDS_BANK_CONFLICT"] * c["launches_pmc_sq2"] for c in sq) / lds) if lds else 0.0
        ent["valu_issue_frac"] = sum(c["SQ_INSTS_VALU"] * c["launches_pmc_sq"] for c in sq) * 2.0 / (NUM_CU * 4 * cyc)
        ent["lds_insts_per_step"] = sum(c["SQ_INSTS_LDS"] * c["launches_pmc_sq"] for c in sq) / steps_in_pmc_run
        ent["lds_wait_ratio"] = (sum(c.get("SQ_WAIT_INST_LDS", 0.0) * c["launches_pmc_sq2"] for c in sq) /
                                 max(1.0, sum(c.get("SQ_WAVE_CYCLES", 0.0) * c["launches_pmc_sq"] for c in sq)))
        # shader clock held during the stage: the counter pass's own cycles over its own kernel durations
        try:
            tr = newest(os.path.join(src, "pmc_sq", "*", "*_kernel_trace.csv"))
            dur = collections.defaultdict(float)
            for r in csv.DictReader(open(tr)):
                dur[r["Kernel_Name"].split("(")[0][:70]] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
            t = sum(dur[k] for k, c in dom.items() if c in sq)
            if t > 0:
                ent["shader_clock_ghz"] = cyc / t
        except Exception:
            pass
    rec["entries"][key] = ent
    json.dump(rec, open(path, "w"), indent=1, sort_keys=True)
    alg = bench["roofline"]["algorithmic_bytes_per_launch"]
    print("%s: traffic %.3f GB per step vs algorithmic %.3f GB (x%.2f)" % (key, per_step / 1e9, alg / 1e9, per_step / alg))
print(open(os.path.join(dst, tag + "_kernel_stats.csv")).read()[:1200])
