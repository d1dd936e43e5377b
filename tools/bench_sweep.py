"""Run bench.py with several extra-argument sets and print one summary line each (GPU box).
usage: python tools/bench_sweep.py "<extra args>" ...   (--steps/--warmup inside an argument set override the defaults)"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for extra in sys.argv[1:]:
    args = extra.split()
    base = [] if "--steps" in args else ["--steps", "5", "--warmup", "2"]
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--quick"] + base + args
    out = subprocess.run(cmd, capture_output=True, text=True).stdout
    line = [l for l in out.splitlines() if l.startswith("{")]
    if not line:
        print(extra, "-> no JSON line"); continue
    d = json.loads(line[-1])
    print("%-72s %8.1f Msamples/s %9.3f ms  frac %.4f  stages %s  plan %s" % (extra, d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["stage_ms"], d["roofline"]["plan"]), flush=True)
