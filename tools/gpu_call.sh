mkdir -p gpurun_out
rm -f gpurun_out/parity_errors_*.json
( timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/pytest_gpu_r2_final.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu_r2_final.log )
tail -3 gpurun_out/pytest_gpu_r2_final.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
ROUND=r2 timeout 1300 bash tools/capture_profiles.sh > gpurun_out/capture.log 2>&1; echo "capture rc=$?"
