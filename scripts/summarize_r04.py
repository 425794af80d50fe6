"""gpurun_out/prof_<tag>/ (scripts/profile_r04.sh) -> profiles/<tag>_*: the BL-5 shard on the time-parallel chunked passes (kernel
stats, calibrated HBM traffic, matrix-pipe counters) and the BL-2 pass pair over three rotating buffer sets (kernel stats,
traffic).  Calibration factors: those of the same run's profiles/<tag>_traffic.json (scripts/summarize_profile.py)."""
import csv, glob, json, os, shutil, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(ROOT, "gpurun_out", f"prof_{tag}"), os.path.join(ROOT, "profiles")
cal = json.load(open(os.path.join(dst, f"{tag}_traffic.json")))["calibration"]


def newest(pattern):
    f = glob.glob(os.path.join(src, pattern), recursive=True)
    return max(f, key=os.path.getmtime) if f else None


def short(name):
    return name.split("(ocs::")[0].split("(int")[0].replace("void ", "").replace("ocs::", "").replace(" ", "")


def counters(folder, names, skip_first=0):
    f = newest(f"{folder}/**/*counter_collection.csv")
    acc = {}
    if not f:
        return acc
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Dispatch_Id"]))
    for row in rows:
        if row["Counter_Name"] in names:
            acc.setdefault(short(row["Kernel_Name"]), {}).setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
    return {k: {n: sum(v[skip_first:]) / max(1, len(v[skip_first:])) for n, v in c.items()} for k, c in acc.items()}


def stats(folder, outname):
    f = newest(f"{folder}/**/*kernel_stats.csv")
    res = {}
    if f:
        with open(f) as fh, open(os.path.join(dst, outname), "w") as out:
            for i, ln in enumerate(fh):
                if i < 24:
                    out.write(ln)
        for row in csv.DictReader(open(f)):
            res[short(row["Name"])] = {"calls": int(row["Calls"]), "avg_us": float(row["AverageNs"]) / 1e3}
    return res


out = {"tag": tag, "calibration": cal}
# ---- BL-5 shard
st = stats("bl5s_trace", f"{tag}_bl5_shard_kernel_stats.csv")
fe, wr = counters("bl5s_fetch", ["FETCH_SIZE"]), counters("bl5s_write", ["WRITE_SIZE"])
mf = counters("bl5s_mfma", ["SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "SQ_INSTS_VALU_MFMA_MOPS_F64", "SQ_BUSY_CYCLES"])
sh = {"note": "scripts/lq_time.py BATCH=1024: LQ32, nC = 4, N = 4000 + 4000, time-parallel chunks (k_lq_forward / k_lq_backward<CH>)",
      "kernels": {}}
tot = 0.0
for k in sorted(set(st) | set(fe) | set(wr)):
    if not k.startswith("k_lq"):
        continue
    e = dict(st.get(k, {}))
    f = fe.get(k, {}).get("FETCH_SIZE", 0.0) * 1024.0 * cal["fetch_factor"]
    w = wr.get(k, {}).get("WRITE_SIZE", 0.0) * 1024.0 * cal["write_factor"]
    e.update({"fetch_corrected": f, "write_corrected": w, "hbm_bytes_per_launch": f + w})
    c = mf.get(k, {})
    if c.get("GRBM_GUI_ACTIVE"):
        e["mfma_busy"] = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (c["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)
        e["executed_mfma_flops"] = c.get("SQ_INSTS_VALU_MFMA_MOPS_F64", 0.0) * 512.0
        if e.get("avg_us"):
            e["executed_TFLOPs"] = e["executed_mfma_flops"] / (e["avg_us"] * 1e-6) / 1e12
    sh["kernels"][k] = e
    if "backward" in k:
        tot += f + w
sh["hbm_bytes_adjoint_kernels_per_pass_pair"] = tot
json.dump(sh, open(os.path.join(dst, f"{tag}_bl5_shard_counters.json"), "w"), indent=1)
json.dump({"batch": 1024, "hbm_bytes_adjoint_kernels_per_pass_pair": tot, "source": f"profiles/{tag}_bl5_shard_counters.json"},
          open(os.path.join(dst, "bl5_shard_traffic_latest.json"), "w"), indent=1)
out["bl5_shard"] = sh
# ---- rotating buffer sets
st = stats("rot_trace", f"{tag}_pair_rotating_kernel_stats.csv")
fe, wr = counters("rot_fetch", ["FETCH_SIZE"], 40), counters("rot_write", ["WRITE_SIZE"], 40)
rot = {"note": "scripts/pair_rotate.py ROTATE=3: BL-2 pass pair, three buffer sets round-robin (steady state: first 40 dispatches dropped)",
       "kernels": {}}
for k in sorted(set(fe) | set(wr)):
    if not (k.startswith("k_forward") or k.startswith("k_backward")):
        continue
    f = fe.get(k, {}).get("FETCH_SIZE", 0.0) * 1024.0 * cal["fetch_factor"]
    w = wr.get(k, {}).get("WRITE_SIZE", 0.0) * 1024.0 * cal["write_factor"]
    rot["kernels"][k] = {**st.get(k, {}), "fetch_corrected": f, "write_corrected": w, "hbm_bytes_per_launch": f + w}
json.dump(rot, open(os.path.join(dst, f"{tag}_pair_rotating_traffic.json"), "w"), indent=1)
out["rotating"] = rot
print(json.dumps(out, indent=1)[:5000])
