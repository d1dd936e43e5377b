import subprocess, signal, time, numpy as np, os, sys
ROOT="/root/repo"
out="/tmp/partial.pfm"
p = subprocess.Popen([os.path.join(ROOT,"simple-spectral"), "-s=cornell", "-w=2048", "-h=2048", "-spp=4096", "--tile-major", "-o=" + out], cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
time.sleep(float(sys.argv[1]) if len(sys.argv)>1 else 4.0)
p.send_signal(signal.SIGINT)
so, se = p.communicate(timeout=120)
print("rc", p.returncode); print(se[-600:]); print(so[-200:])
data = np.fromfile(out, dtype="<f4", offset=len("PF\n2048 2048\n-1.0\n")).reshape(2048, 2048, 3)
for r in (0, 8, 512, 1024, 1536, 2040):
    row = data[r:r+8, :, 0]
    print(r, "min %.4f max %.4f uniq %d" % (row.min(), row.max(), len(np.unique(row))), row[0,:4], row[0,8:12])
