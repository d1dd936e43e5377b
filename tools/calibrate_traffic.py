"""FETCH_SIZE / WRITE_SIZE calibration: counter bytes per true byte for each access pattern of
tools/ubench/traffic_calib (rocprofv3 --pmc passes collected by tools/profile_round.sh <tag> calib).
Writes profiles/<name>/traffic_calibration.json.     usage: calibrate_traffic.py <tag> <profiles subdir>"""
import csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, name = sys.argv[1], sys.argv[2]
G = os.path.join(ROOT, "gpurun_out")


def counter_kib(d, counter):
    f = glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True)
    if not f:
        return None
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f[0])) if r["Counter_Name"] == counter and "calib_" in r["Kernel_Name"]]
    return sum(v) / len(v) if v else None


out = {}
for p in ("rd16", "rd8", "wr16", "wr8", "rmw", "rd16s"):
    j = json.loads(open(os.path.join(G, "%s_calib_%s.json" % (tag, p))).read().strip().splitlines()[-1])
    f = counter_kib(os.path.join(G, "%s_calib_%s_fetch" % (tag, p)), "FETCH_SIZE")
    w = counter_kib(os.path.join(G, "%s_calib_%s_write" % (tag, p)), "WRITE_SIZE")
    out[p] = {"true_read_bytes": j["read_bytes"], "true_write_bytes": j["write_bytes"], "FETCH_SIZE_KiB": f, "WRITE_SIZE_KiB": w,
              "fetch_per_true_read": (f * 1024 / j["read_bytes"]) if f is not None and j["read_bytes"] else None,
              "write_per_true_write": (w * 1024 / j["write_bytes"]) if w is not None and j["write_bytes"] else None}
    print(p, out[p])
os.makedirs(os.path.join(ROOT, "profiles", name), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "profiles", name, "traffic_calibration.json"), "w"), indent=1, sort_keys=True)
