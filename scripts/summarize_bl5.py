"""gpurun_out/prof_<tag>_bl5/ (scripts/profile_bl5.sh) -> profiles/<tag>_bl5_counters.json, <tag>_bl5_kernel_stats.csv and
bl5_traffic_latest.json (which bench.py replays).  Calibration of FETCH / WRITE as in profiles/r03_traffic.json."""
import csv, glob, json, os, shutil, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r03e"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(ROOT, "gpurun_out", f"prof_{tag}_bl5"), os.path.join(ROOT, "profiles")
cal = json.load(open(os.path.join(dst, "r03_traffic.json")))["calibration"]
def newest(pattern):
    f = glob.glob(os.path.join(src, pattern), recursive=True)
    return max(f, key=os.path.getmtime) if f else None
def short(name):
    return name.split("(ocs::")[0].replace("void ", "").replace("ocs::", "").replace(" ", "")
def counters(folder):
    acc = {}
    f = newest(f"{folder}/**/*counter_collection.csv")
    for row in csv.DictReader(open(f)) if f else []:
        acc.setdefault(short(row["Kernel_Name"]), {}).setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
    return {k: {n: sum(v) / len(v) for n, v in c.items()} for k, c in acc.items()}
st = {}
f = newest("trace/**/*kernel_stats.csv")
if f:
    shutil.copy(f, os.path.join(dst, f"{tag}_bl5_kernel_stats.csv"))
    for row in csv.DictReader(open(f)):
        st[short(row["Name"])] = {"calls": int(row["Calls"]), "avg_us": float(row["AverageNs"]) / 1e3}
fe, wr, mf, va = counters("fetch"), counters("write"), counters("mfma"), counters("valu")
out = {"tag": tag, "calibration": cal, "kernels": {},
       "note": "scripts/lq_time.py: LQ32, nC = 4, N = 4000 + 4000, batch 8192 (two-wave kernels k_lq2_*); mean per dispatch; "
               "SQ cycle counters in quad-cycles summed over waves"}
tot = 0.0
for k in sorted(set(fe) | set(wr) | set(mf)):
    if "k_lq" not in k:
        continue
    e = {}
    fb, wb = fe.get(k, {}).get("FETCH_SIZE", 0.0) * 1024 * cal["fetch_factor"], wr.get(k, {}).get("WRITE_SIZE", 0.0) * 1024 * cal["write_factor"]
    e.update({"fetch_corrected": fb, "write_corrected": wb, "hbm_bytes_per_launch": fb + wb})
    tot += fb + wb
    c = mf.get(k, {})
    if c.get("GRBM_GUI_ACTIVE"):
        e["mfma_busy"] = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (c["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)   # busy summed over SIMDs, GUI_ACTIVE over the 8 XCDs
    e["executed_mfma_flops"] = c.get("SQ_INSTS_VALU_MFMA_MOPS_F64", 0.0) * 512.0
    e["counters"] = {**c, **va.get(k, {})}
    v = va.get(k, {})
    if v.get("SQ_WAVES"):
        e["valu_insts_per_wave"] = v.get("SQ_INSTS_VALU", 0.0) / v["SQ_WAVES"]
        if v.get("SQ_WAVE_CYCLES"):
            e["share_active_valu_incl_mfma"] = v.get("SQ_ACTIVE_INST_VALU", 0.0) / v["SQ_WAVE_CYCLES"]
            e["share_wait_any"] = v.get("SQ_WAIT_ANY", 0.0) / v["SQ_WAVE_CYCLES"]
    if k in st:
        e["avg_us"] = st[k]["avg_us"]
        e["executed_TFLOPs"] = e["executed_mfma_flops"] / (st[k]["avg_us"] * 1e-6) / 1e12
    out["kernels"][k] = e
out["hbm_bytes_per_pass_pair"] = tot
json.dump(out, open(os.path.join(dst, f"{tag}_bl5_counters.json"), "w"), indent=1)
json.dump({"batch": 8192, "hbm_bytes_per_pass_pair": tot, "kernels": {k: {"hbm_bytes_per_launch": e["hbm_bytes_per_launch"]} for k, e in out["kernels"].items()},
           "source": f"profiles/{tag}_bl5_counters.json"}, open(os.path.join(dst, "bl5_traffic_latest.json"), "w"), indent=1)
for k, e in out["kernels"].items():
    print(k, {n: (round(v, 3) if isinstance(v, float) else v) for n, v in e.items() if n != "counters"})
