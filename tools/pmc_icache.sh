#!/bin/bash
# Instruction-cache and wait counters of the path kernel (two rocprofv3 --pmc passes of bench.py): the hot loop is ~50 KB of code
# on a 64 KB instruction cache shared by two CUs -- does it miss?  (r03: 2.8 k misses in 3.47 G requests: no.)
R=$(pwd); cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o -i "SQC_[A-Z_0-9]*\|SQ_IFETCH[A-Z_0-9]*\|SQ_INST_CYCLES[A-Z_0-9]*\|SQ_WAIT[A-Z_0-9]*\|SQ_BUSY[A-Z_0-9]*\|SQ_INSTS_[A-Z_0-9]*\|SQ_VALU[A-Z_0-9]*\|SQ_ACTIVE[A-Z_0-9]*" | sort -u > $R/gpurun_out/pmc_avail.txt
# (a third pass with nine SQ_INST_CYCLES_* / SQ_ACTIVE_INST_* counters at once never returned on this pool: every pass runs under `timeout`)
for SET in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
D=$R/gpurun_out/pmc_ic_$(echo $SET | cut -c1-12 | tr ' ' _)
rm -rf $D
timeout 180 rocprofv3 --pmc $SET --output-format csv -d $D -- python $R/bench.py --steps 2 --warmup 1 --quick > /dev/null 2>&1
python3 - $D <<'PY'
import csv, glob, sys, collections
rows = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows[(r["Kernel_Name"].split("(")[0][:40], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(rows.items()):
    if k.startswith("ssx_render"):
        print(k, c, len(v), "%.5g" % (sum(v) / len(v)))
PY
done
