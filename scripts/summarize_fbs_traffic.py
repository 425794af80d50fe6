"""gpurun_out/fbstraffic_<tag>/ (scripts/profile_fbs_traffic.sh) -> profiles/<tag>_fb_sweep_traffic.json and
profiles/fb_traffic_latest.json: calibrated HBM bytes per launch of the two kernels of a folded sweep (live launches only:
a launch that finds its gate closed moves nothing and is left out)."""
import csv, glob, json, os, sys
tag = sys.argv[1]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", f"fbstraffic_{tag}")
cal = json.load(open(os.path.join(ROOT, "profiles", f"{tag}_traffic.json")))["calibration"]


def per_kernel(folder, counter):
    f = max(glob.glob(os.path.join(src, folder, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)
    acc = {}
    for row in csv.DictReader(open(f)):
        if row["Counter_Name"] != counter:
            continue
        n = row["Kernel_Name"]
        key = "k_forward_cc" if "k_forward_cc" in n else "k_costate_plx" if "k_costate_plx" in n else None
        if key:
            acc.setdefault(key, []).append(float(row["Counter_Value"]) * 1024.0)
    out = {}
    for k, v in acc.items():
        live = [x for x in v if x > 0.05 * max(v)]
        out[k] = (sum(live) / len(live), len(live), len(v))
    return out


fe, wr = per_kernel("pmc_fetch", "FETCH_SIZE"), per_kernel("pmc_write", "WRITE_SIZE")
out = {"tag": tag, "units": "bytes per live launch (batch 16384, N = 1000)", "calibration": cal, "kernels": {}}
tot = 0.0
for k in ("k_forward_cc", "k_costate_plx"):
    f = fe[k][0] * cal["fetch_factor"]
    w = wr[k][0] * cal["write_factor"]
    out["kernels"][k] = {"fetch_corrected": f, "write_corrected": w, "hbm_bytes_per_launch": f + w,
                         "live_dispatches": fe[k][1], "dispatches": fe[k][2]}
    tot += f + w
out["hbm_bytes_per_batch_sweep"] = tot
out["algorithmic_bytes_per_batch_sweep"] = 40.0 * 16384 * 1000
for name in (f"{tag}_fb_sweep_traffic.json", "fb_traffic_latest.json"):
    json.dump(out, open(os.path.join(ROOT, "profiles", name), "w"), indent=1)
print(json.dumps(out, indent=1))
