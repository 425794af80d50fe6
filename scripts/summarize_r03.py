"""gpurun_out/prof_<tag>/ (scripts/profile_r03.sh) -> profiles/<tag>_*: kernel statistics and calibrated HBM traffic of the BL-4,
BL-5 and fb_sweep kernels, matrix-pipe counters of BL-5.  Calibration factors (true / reported bytes of an 8 B-per-lane copy,
MI355X_MICROARCH.md section HBM) are those of the same run's profiles/<tag>_traffic.json (scripts/summarize_profile.py)."""
import csv, glob, json, os, shutil, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(ROOT, "gpurun_out", f"prof_{tag}"), os.path.join(ROOT, "profiles")
cal = json.load(open(os.path.join(dst, f"{tag}_traffic.json")))["calibration"]


def newest(pattern):
    f = glob.glob(os.path.join(src, pattern), recursive=True)
    return max(f, key=os.path.getmtime) if f else None


def short(name):
    return name.split("(ocs::")[0].replace("void ", "").replace("ocs::", "").replace(" ", "")


def counters(folder, names):
    """mean over the LIVE dispatches of each kernel (a launch that finds its gate closed moves nothing)"""
    f = newest(f"{folder}/**/*counter_collection.csv")
    acc = {}
    if not f:
        return acc
    for row in csv.DictReader(open(f)):
        if row["Counter_Name"] in names:
            acc.setdefault(short(row["Kernel_Name"]), {}).setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
    out = {}
    for k, c in acc.items():
        out[k] = {}
        for n, v in c.items():
            live = [x for x in v if x > 0.05 * max(v)] if max(v) > 0 else v
            out[k][n] = sum(live) / len(live)
    return out


def stats(folder, outname):
    f = newest(f"{folder}/**/*kernel_stats.csv")
    res = {}
    if f:
        shutil.copy(f, os.path.join(dst, outname))
        for row in csv.DictReader(open(f)):
            res[short(row["Name"])] = {"calls": int(row["Calls"]), "avg_us": float(row["AverageNs"]) / 1e3, "max_us": float(row["MaxNs"]) / 1e3}
    return res


def traffic(prefix, want):
    fe, wr = counters(prefix + "_fetch", ["FETCH_SIZE"]), counters(prefix + "_write", ["WRITE_SIZE"])
    out, tot = {}, 0.0
    for k in sorted(set(fe) | set(wr)):
        if not any(w in k for w in want):
            continue
        f = fe.get(k, {}).get("FETCH_SIZE", 0.0) * 1024.0 * cal["fetch_factor"]
        w = wr.get(k, {}).get("WRITE_SIZE", 0.0) * 1024.0 * cal["write_factor"]
        out[k] = {"fetch_corrected": f, "write_corrected": w, "hbm_bytes_per_launch": f + w}
        tot += f + w
    return out, tot


summary = {"tag": tag, "calibration": cal, "note": "bytes per live launch; FETCH_SIZE / WRITE_SIZE in separate rocprofv3 --pmc passes"}
# ---- BL-4
bl4 = {"entries": {}}
for B in (8192, 65536):
    st = stats(f"bl4_{B}_trace", f"{tag}_bl4_{B}_kernel_stats.csv")
    tr, tot = traffic(f"bl4_{B}", ["k_forward_p2", "k_backward_fcs", "k_forward_fc", "k_backward_fc"])
    alg = 16.0 * B * 1000
    bl4["entries"][str(B)] = {"batch": B, "kernels": tr, "kernel_stats": {k: v for k, v in st.items() if k.startswith("k_")},
                              "hbm_bytes_per_evaluation": tot, "algorithmic_bytes_checkpoint_write_plus_read": alg,
                              "source": f"profiles/{tag}_bl4_traffic.json"}
json.dump(bl4, open(os.path.join(dst, f"{tag}_bl4_traffic.json"), "w"), indent=1)
json.dump(bl4, open(os.path.join(dst, "bl4_traffic_latest.json"), "w"), indent=1)
summary["bl4"] = bl4["entries"]
# ---- BL-5
st5 = stats("bl5_trace", f"{tag}_bl5_kernel_stats.csv")
tr5, tot5 = traffic("bl5", ["k_lq"])
mf = counters("bl5_mfma", ["SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "SQ_INSTS_VALU_MFMA_MOPS_F64", "SQ_BUSY_CYCLES"])
bl5 = {"kernels": {}}
for k, c in mf.items():
    if "k_lq" not in k:
        continue
    e = {"counters": c}
    if c.get("GRBM_GUI_ACTIVE"):
        # BUSY_CYCLES summed over the SIMDs; GUI_ACTIVE summed over the 8 XCDs (MI355X_MICROARCH.md, DVFS)
        e["mfma_busy"] = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (c["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)
    e["executed_mfma_flops"] = c.get("SQ_INSTS_VALU_MFMA_MOPS_F64", 0.0) * 512.0
    if k in st5:
        e["avg_us_untraced_counters_off"] = st5[k]["avg_us"]
        e["executed_TFLOPs"] = e["executed_mfma_flops"] / (st5[k]["avg_us"] * 1e-6) / 1e12
    if k in tr5:
        e.update(tr5[k])
    bl5["kernels"][k] = e
bl5["hbm_bytes_per_pass_pair"] = tot5
bl5["note"] = "scripts/lq_time.py: LQ32, nC = 4, N = 4000 + 4000, batch 8192 (two-wave kernels k_lq2_*)"
json.dump(bl5, open(os.path.join(dst, f"{tag}_bl5_counters.json"), "w"), indent=1)
json.dump({"batch": 8192, "hbm_bytes_per_pass_pair": tot5, "kernels": tr5, "source": f"profiles/{tag}_bl5_counters.json"},
          open(os.path.join(dst, "bl5_traffic_latest.json"), "w"), indent=1)
summary["bl5"] = bl5
# ---- fb_sweep
stf = stats("fbs_trace", f"{tag}_fb_sweep_kernel_stats.csv")
trf, totf = traffic("fbs", ["k_forward_cc", "k_costate"])
fb = {"tag": tag, "units": "bytes per live launch (batch 16384, N = 1000)", "calibration": cal, "kernels": trf,
      "hbm_bytes_per_batch_sweep": totf, "algorithmic_bytes_per_batch_sweep": 40.0 * 16384 * 1000}
json.dump(fb, open(os.path.join(dst, f"{tag}_fb_sweep_traffic.json"), "w"), indent=1)
json.dump(fb, open(os.path.join(dst, "fb_traffic_latest.json"), "w"), indent=1)
summary["fb_sweep"] = {"traffic": fb, "kernel_stats": {k: v for k, v in stf.items() if k.startswith("k_")}}
print(json.dumps(summary, indent=1)[:6000])
