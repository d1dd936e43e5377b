"""Run bench.py with several extra-argument sets and print one summary line each (GPU box)."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for extra in sys.argv[1:]:
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "5", "--warmup", "2", "--no-cpu-baseline"] + extra.split()
    out = subprocess.run(cmd, capture_output=True, text=True).stdout
    line = [l for l in out.splitlines() if l.startswith("{")]
    if not line:
        print(extra, "-> no JSON line"); continue
    d = json.loads(line[-1])
    print("%-28s %8.1f Msamples/s %8.3f ms  frac %.4f  stages %s" % (extra, d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["stage_ms"]))
