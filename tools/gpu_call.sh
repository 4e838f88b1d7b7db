mkdir -p gpurun_out
rm -f gpurun_out/parity_errors_*.json
( timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/pytest_gpu_r2f.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu_r2f.log )
tail -4 gpurun_out/pytest_gpu_r2f.log
timeout 900 python tools/bench_configs.py > gpurun_out/configs.log 2>&1; echo "configs rc=$?"
ROUND=r2 timeout 1300 bash tools/capture_profiles.sh > gpurun_out/capture.log 2>&1; echo "capture rc=$?"
