"""gpurun_out/prof_<tag>v/ (scripts/profile_variants.sh) -> profiles/<tag>_bl2_variants_traffic.json (+ bl2_variants_traffic_latest.json,
which bench.py replays): calibrated HBM bytes per pass pair of the lane mapping at batch 65536 and of the one-state shapes.
Calibration as in profiles/<tag>_traffic.json (8 B-per-lane copy, MI355X_MICROARCH.md section HBM)."""
import csv, glob, json, os, shutil, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(ROOT, "gpurun_out", f"prof_{tag}v"), os.path.join(ROOT, "profiles")
cal = json.load(open(os.path.join(dst, f"{tag}_traffic.json")))["calibration"]
def newest(pattern):
    f = glob.glob(os.path.join(src, pattern), recursive=True)
    return max(f, key=os.path.getmtime) if f else None
def short(name):
    return name.split("(ocs::")[0].replace("void ", "").replace("ocs::", "").replace(" ", "")
def counter(folder, cname):
    acc = {}
    f = newest(f"{folder}/**/*counter_collection.csv")
    for row in csv.DictReader(open(f)) if f else []:
        if row["Counter_Name"] == cname:
            acc.setdefault(short(row["Kernel_Name"]), []).append(float(row["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}
out = {"tag": tag, "calibration": cal, "entries": {}}
for key, pre, alg in (("lane_4_65536", "lane65536", 8.0 * (3 * 5 + 6) * 65536 * 1000), ("auto_1_4096", "ns1", 8.0 * (3 * 2 + 6) * 4096 * 1000)):
    fe, wr = counter(pre + "_fetch", "FETCH_SIZE"), counter(pre + "_write", "WRITE_SIZE")
    ker, tot = {}, 0.0
    for k in sorted(set(fe) | set(wr)):
        if not k.startswith(("k_forward", "k_backward")):
            continue
        f, w = fe.get(k, 0.0) * 1024.0 * cal["fetch_factor"], wr.get(k, 0.0) * 1024.0 * cal["write_factor"]
        ker[k] = {"fetch_corrected": f, "write_corrected": w, "hbm_bytes_per_launch": f + w}
        tot += f + w
    st = newest(f"{pre}_trace/**/*kernel_stats.csv")
    if st:
        shutil.copy(st, os.path.join(dst, f"{tag}_{pre}_kernel_stats.csv"))
    out["entries"][key] = {"kernels": ker, "hbm_bytes_per_pass_pair": tot, "algorithmic_bytes_per_pass_pair": alg,
                           "source": f"profiles/{tag}_bl2_variants_traffic.json"}
    print(key, f"{tot/1e6:.1f} MB measured, {alg/1e6:.1f} MB algorithmic", {k: round(v['hbm_bytes_per_launch'] / 1e6, 1) for k, v in ker.items()})
json.dump(out, open(os.path.join(dst, f"{tag}_bl2_variants_traffic.json"), "w"), indent=1)
json.dump(out, open(os.path.join(dst, "bl2_variants_traffic_latest.json"), "w"), indent=1)
