"""Turns gpurun_out/prof_<tag>/ (scripts/profile_round.sh) into the committed evidence under profiles/:
  profiles/<tag>_kernel_stats.csv      rocprofv3 --kernel-trace --stats summary of bench.py
  profiles/<tag>_traffic.json          HBM bytes per launch of the RK4 kernels from FETCH_SIZE / WRITE_SIZE,
                                       with the calibration factors measured on a known byte count
  profiles/traffic_latest.json         what bench.py reports as roofline.traffic
"""
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)


def one(pattern):
    f = glob.glob(os.path.join(src, pattern), recursive=True)
    if not f:
        raise SystemExit(f"missing {pattern} under {src}")
    return max(f, key=os.path.getmtime)  # gpurun merges runs: take the newest


shutil.copy(one("trace/**/*kernel_stats.csv"), os.path.join(dst, f"{tag}_kernel_stats.csv"))


def counter_per_kernel(folder, counter):
    """mean counter value per dispatch, keyed by a short kernel name"""
    acc = {}
    with open(one(f"{folder}/**/*counter_collection.csv")) as fh:
        for row in csv.DictReader(fh):
            if row["Counter_Name"] != counter:
                continue
            name = row["Kernel_Name"]
            key = "k_forward" if "k_forward" in name else "k_backward" if "k_backward" in name else \
                  "k_fill" if "k_fill_lam_cost_row" in name else "k_copy8" if "k_copy8" in name else None
            if key:
                acc.setdefault(key, []).append(float(row["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}, {k: len(v) for k, v in acc.items()}


fetch, nf = counter_per_kernel("pmc_fetch", "FETCH_SIZE")
write, nw = counter_per_kernel("pmc_write", "WRITE_SIZE")
cfetch, _ = counter_per_kernel("cal_fetch", "FETCH_SIZE")
cwrite, _ = counter_per_kernel("cal_write", "WRITE_SIZE")
cal_bytes = 5 * 1001 * 65536 * 8
# counters are in KiB; calibration factor = true bytes / reported bytes for 8 B/lane streams
kf = cal_bytes / (cfetch["k_copy8"] * 1024.0)
kw = cal_bytes / (cwrite["k_copy8"] * 1024.0)
stats = {}
with open(os.path.join(dst, f"{tag}_kernel_stats.csv")) as fh:
    for row in csv.DictReader(fh):
        for key in ("k_forward", "k_backward"):
            if key in row["Name"]:
                stats[key] = {"calls": int(row["Calls"]), "avg_ns": float(row["AverageNs"])}
out = {"tag": tag, "units": "bytes per launch", "counter_unit": "KiB (rocprofv3 FETCH_SIZE / WRITE_SIZE)",
       "calibration": {"known_bytes_each_way": cal_bytes, "fetch_factor": kf, "write_factor": kw,
                       "note": "ocs_copy_dev, 8 B/lane straight copy; factor = true / reported"},
       "kernels": {}}
for key in ("k_forward", "k_backward"):
    raw_f, raw_w = fetch[key] * 1024.0, write[key] * 1024.0
    out["kernels"][key] = {"fetch_raw": raw_f, "write_raw": raw_w, "fetch_corrected": raw_f * kf,
                           "write_corrected": raw_w * kw, "hbm_bytes_per_launch": raw_f * kf + raw_w * kw,
                           "dispatches_sampled": nf[key], **stats.get(key, {})}
# bench.py times compute_adjoints as one unit: the wave-specialised backward kernel plus the streaming fill of
# the constant cost row of lam; report their traffic together as the backward launch
bwd_total = out["kernels"]["k_backward"]["hbm_bytes_per_launch"]
if "k_fill" in write:
    fill = fetch.get("k_fill", 0.0) * 1024.0 * kf + write["k_fill"] * 1024.0 * kw
    out["kernels"]["k_fill_lam_cost_row"] = {"hbm_bytes_per_launch": fill}
    bwd_total += fill
out["backward_unit_hbm_bytes"] = bwd_total
json.dump(out, open(os.path.join(dst, f"{tag}_traffic.json"), "w"), indent=1)
json.dump({"kernel": "k_backward", "batch": 4096, "hbm_bytes_per_launch": bwd_total,
           "source": f"profiles/{tag}_traffic.json"}, open(os.path.join(dst, "traffic_latest.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
