"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, as the MI355X guide
prescribes) of `bench.py` into profiles/traffic.json: HBM bytes per launch of ssx_render_kernel.
Corrections per /opt/skills/guides/MI355X_MICROARCH.md "HBM": both counters are in KiB; on gfx950
FETCH_SIZE reports half the bytes of a wide (16 B/lane) coalesced read stream -- the record and
frame reads of this pipeline are exactly that -- so it is doubled; WRITE_SIZE is taken as is
(uncalibrated).   usage: collect_traffic.py <fetch_dir> <write_dir> "<key>" """
import csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

def mean_counter(d, name, kernel="ssx_render_kernel"):
    f = glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True)[0]
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if r["Counter_Name"] == name and kernel in r["Kernel_Name"]]
    return sum(v) / len(v)

fetch_kib = mean_counter(sys.argv[1], "FETCH_SIZE")
write_kib = mean_counter(sys.argv[2], "WRITE_SIZE")
total = int(2 * fetch_kib * 1024 + write_kib * 1024)
path = os.path.join(ROOT, "profiles", "traffic.json")
t = json.load(open(path)) if os.path.exists(path) else {}
t[sys.argv[3]] = total
sys.path.insert(0, ROOT)
import bench  # kernel_source_id(): the hash of the kernel sources these counters belong to (bench.py marks a replayed figure stale when they change)
gen_f, gen_w = mean_counter(sys.argv[1], "FETCH_SIZE", "ssx_generate_kernel"), mean_counter(sys.argv[2], "WRITE_SIZE", "ssx_generate_kernel")
t[sys.argv[3] + " detail"] = {"FETCH_SIZE_KiB": {"path": fetch_kib, "generate": gen_f}, "WRITE_SIZE_KiB": {"path": write_kib, "generate": gen_w}, "fetch_correction": 2.0,
                              "bytes_per_launch": {"path": total, "generate": int(2 * gen_f * 1024 + gen_w * 1024)}, "kernel_source_id": bench.kernel_source_id(),
                              "taken_by": "tools/profile_round.sh + tools/collect_traffic.py (rocprofv3 --pmc, separate passes)"}
json.dump(t, open(path, "w"), indent=1, sort_keys=True)
print(sys.argv[3], "->", total, "bytes per launch")
