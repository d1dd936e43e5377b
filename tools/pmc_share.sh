#!/bin/bash
# SQ counters of the path kernel for one rank's share of the headline config (tools/rank_share.py --only-n N): what differs between
# the shares of N = 1 and N = 8 -- instructions, lane cycles, waits?   usage: tools/pmc_share.sh N [N ...]
R=$(pwd); cd /tmp && export TMPDIR=/tmp
for N in "$@"; do
for SET in "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_ANY" "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
	D=$R/gpurun_out/pmcshare_$N; rm -rf $D
	timeout 300 rocprofv3 --pmc $SET --output-format csv -d $D -- python $R/tools/rank_share.py --configs headline --only-n $N --steps 2 --warmup 1 > /dev/null 2>&1
	python3 - $D $N <<'PY'
import csv, glob, sys, collections
rows = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows[(r["Kernel_Name"].split("(")[0][:40], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(rows.items()):
    if k.startswith("ssx_render"):
        print("N=%s" % sys.argv[2], k, c, len(v), "%.5g" % (sum(v) / len(v)))
PY
done; done
