#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace stats + separate PMC passes of the
# default bench.py command, into gpurun_out/<tag>_*, and the FETCH_SIZE / WRITE_SIZE calibration
# passes over tools/ubench/traffic_calib (known byte counts in the pipeline's access patterns).
# Counter passes carry no trace flags.
# usage: tools/profile_round.sh <tag> [calib]   then locally: python tools/summarize_profile.py <tag> ...
TAG=${1:-final}
R=$(pwd)
B="python $R/bench.py --steps 4 --warmup 1 --quick $BENCH_ARGS" # BENCH_ARGS="--scene plane-srgb --res 1024 --spp 1024": another workload
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_stats -- $B > $O/${TAG}_stats.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_ANY --output-format csv -d $O/${TAG}_sq1 -- $B > /dev/null 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/${TAG}_sq2 -- $B > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/${TAG}_fetch -- $B > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/${TAG}_write -- $B > /dev/null 2>&1
if [ "$2" = "calib" ]; then
	for P in rd16 rd8 wr16 wr8 rmw rd16s; do
		rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/${TAG}_calib_${P}_fetch -- $R/tools/ubench/traffic_calib $P > $O/${TAG}_calib_${P}.json 2>/dev/null
		rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/${TAG}_calib_${P}_write -- $R/tools/ubench/traffic_calib $P > /dev/null 2>&1
	done
fi
cut -c1-150 $O/${TAG}_stats/*/*kernel_stats.csv | head -8
tail -1 $O/${TAG}_stats.log | cut -c1-400
