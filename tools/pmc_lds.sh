#!/bin/bash
# One rocprofv3 --pmc pass with the LDS counters (bank-conflict cycles against all LDS-array cycles, LDS instructions) of bench.py for the
# product library and for the variant libraries given, per kernel: tools/pmc_lds.sh "<bench args>" [variant.so ...]
export SSX_DEBUG_ENV=1 # the master switch of the A/B environment variables (README)
R=$(pwd); ARGS=$1; shift
cd /tmp && export TMPDIR=/tmp
for V in product "$@"; do
	D=$R/gpurun_out/pmcl_$(basename $V .so)
	rm -rf $D
	if [ "$V" = product ]; then unset SSX_HIP_LIB_OVERRIDE; else export SSX_HIP_LIB_OVERRIDE=$R/$V; fi
	rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_INSTS_VALU --output-format csv -d $D -- python $R/bench.py --steps 2 --warmup 1 --quick $ARGS > /dev/null 2>&1
	python3 - $D $V <<'PY'
import csv, glob, sys, collections
rows = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows[(r["Kernel_Name"].split("(")[0][:40], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(rows.items()):
    if k.startswith("ssx_render") or k.startswith("ssx_generate"):
        big = [x for x in v if x > 0.5 * max(v)] or v
        print(sys.argv[2], k, c, len(big), "%.5g" % (sum(big) / len(big)))
PY
done
