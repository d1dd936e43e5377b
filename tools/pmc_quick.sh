#!/bin/bash
# One rocprofv3 --pmc pass (SQ instruction / cycle counters) of bench.py for the product library and for the variant
# libraries given, printed per kernel: tools/pmc_quick.sh "<bench args>" [variant.so ...]
export SSX_DEBUG_ENV=1 # the master switch of the A/B environment variables (README)
R=$(pwd); ARGS=$1; shift
cd /tmp && export TMPDIR=/tmp
for V in product "$@"; do
	D=$R/gpurun_out/pmcq_$(basename $V .so)
	rm -rf $D
	if [ "$V" = product ]; then unset SSX_HIP_LIB_OVERRIDE; else export SSX_HIP_LIB_OVERRIDE=$R/$V; fi
	rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_ANY --output-format csv -d $D -- python $R/bench.py --steps 2 --warmup 1 --quick $ARGS > /dev/null 2>&1
	python3 - $D $V <<'PY'
import csv, glob, sys, collections
rows = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows[(r["Kernel_Name"].split("(")[0][:40], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(rows.items()):
    if k.startswith("ssx_render") or k.startswith("ssx_generate"):
        print(sys.argv[2], k, c, len(v), "%.4g" % (sum(v) / len(v)))
PY
done
