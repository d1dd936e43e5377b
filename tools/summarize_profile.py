"""Condense the rocprofv3 outputs of tools/profile_round.sh (gpurun_out/<tag>_*) into
profiles/<name>/: kernel_stats.csv (the --stats table as is) and pmc_summary.csv (mean per dispatch
of every counter per kernel), and refresh profiles/traffic.json via collect_traffic.py.
usage: summarize_profile.py <tag> <profiles subdir> "<traffic key>" """
import csv, glob, os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, name, key = sys.argv[1], sys.argv[2], sys.argv[3]
G = os.path.join(ROOT, "gpurun_out")
out = os.path.join(ROOT, "profiles", name)
os.makedirs(out, exist_ok=True)
shutil.copy(glob.glob(os.path.join(G, tag + "_stats", "**", "*kernel_stats.csv"), recursive=True)[0], os.path.join(out, "kernel_stats.csv"))
rows = {}
for part in ("sq1", "sq2", "fetch", "write"):
    for f in glob.glob(os.path.join(G, "%s_%s" % (tag, part), "**", "*_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = (r["Kernel_Name"].split("(")[0][:60], r["Counter_Name"])
            rows.setdefault(k, []).append(float(r["Counter_Value"]))
with open(os.path.join(out, "pmc_summary.csv"), "w") as f:
    f.write("kernel,counter,dispatches,mean_per_dispatch\n")
    for (kern, ctr), v in sorted(rows.items()):
        f.write("%s,%s,%d,%.1f\n" % (kern, ctr, len(v), sum(v) / len(v)))
subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "collect_traffic.py"), os.path.join(G, tag + "_fetch"), os.path.join(G, tag + "_write"), key])
print(open(os.path.join(out, "kernel_stats.csv")).read()[:900])
