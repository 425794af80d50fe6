"""gpurun_out/sq_<tag>/ (scripts/profile_sq.sh) -> profiles/<tag>_sq.json: per kernel, the mean over dispatches of
every collected SQ counter plus the derived shares the design discussion uses.  SQ cycle counters are in
quad-cycles summed over waves (MI355X_MICROARCH.md, cycle constants)."""
import csv, glob, json, os, sys
tag = sys.argv[1]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", f"sq_{tag}")
want = sys.argv[2:] or ["k_forward", "k_backward", "k_costate", "k_control", "k_lq"]
acc = {}
for f in glob.glob(os.path.join(src, "pass*", "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            name = row["Kernel_Name"]
            key = next((w for w in want if w in name), None)
            if not key:
                continue
            short = name.split("(ocs::")[0].replace("void ", "").replace("ocs::", "").replace(" ", "")
            acc.setdefault(short, {}).setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
dur = {}
for f in glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            short = row["Name"].split("(ocs::")[0].replace("void ", "").replace("ocs::", "").replace(" ", "")
            dur[short] = {"calls": int(row["Calls"]), "avg_us": float(row["AverageNs"]) / 1e3}
out = {"tag": tag, "note": "mean per dispatch; SQ_*_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* in quad-cycles summed over waves", "kernels": {}}
for k, c in sorted(acc.items()):
    m = {n: sum(v) / len(v) for n, v in c.items()}
    d = dict(m)
    wc = m.get("SQ_WAVE_CYCLES")
    if wc:
        for n in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU",
                  "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_MISC"):
            if n in m:
                d["share_" + n[3:].lower()] = m[n] / wc
    if m.get("SQ_WAVES"):
        for n in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_SMEM", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR"):
            if n in m:
                d["per_wave_" + n[3:].lower()] = m[n] / m["SQ_WAVES"]
        if wc:
            d["wave_cycles_per_wave_x4"] = 4 * wc / m["SQ_WAVES"]
    if k in dur:
        d["duration"] = dur[k]
    out["kernels"][k] = d
os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
dst = os.path.join(ROOT, "profiles", f"{tag}_sq.json")
json.dump(out, open(dst, "w"), indent=1)
print("wrote", dst)
for k, d in out["kernels"].items():
    print(k, {n: (round(v, 3) if isinstance(v, float) else v) for n, v in d.items() if n.startswith(("share_", "per_wave", "wave_cyc", "duration"))})
