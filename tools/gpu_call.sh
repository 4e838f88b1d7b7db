mkdir -p gpurun_out
rm -f gpurun_out/parity_errors_*.json
( timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=25 > gpurun_out/pytest_gpu_r2b.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu_r2b.log )
tail -45 gpurun_out/pytest_gpu_r2b.log
timeout 200 python tools/copy_probe.py > gpurun_out/copy_probe.log 2>&1; tail -30 gpurun_out/copy_probe.log
timeout 300 python tools/hub_attention_probe.py > gpurun_out/hub_attention_probe.log 2>&1; tail -5 gpurun_out/hub_attention_probe.log
ROUND=r2 timeout 1300 bash tools/capture_profiles.sh > gpurun_out/capture.log 2>&1; tail -5 gpurun_out/capture.log
timeout 600 python tools/emulate_sharded.py --world 8 --single-gpu-ms 6.98 --link-gbps 61 76.8 > gpurun_out/emulate_r2b.log 2>&1; echo "emulate rc=$?"; grep -c step_ms gpurun_out/emulate_r2b.log
