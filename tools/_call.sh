cd $GRAFT_REPO_ROOT
bash tools/sanitize.sh --gpu-only gpurun_out/r05_sanitize_gpu.log > /dev/null 2>&1; echo "sanitize rc=$?"; cat gpurun_out/r05_sanitize_gpu.log
