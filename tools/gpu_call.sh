mkdir -p gpurun_out
rm -f gpurun_out/parity_errors_*.json
( timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=8 > gpurun_out/pytest_gpu_r2d.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu_r2d.log )
tail -25 gpurun_out/pytest_gpu_r2d.log
timeout 300 python tools/hub_attention_probe.py > gpurun_out/hub_attention_probe.log 2>&1; tail -4 gpurun_out/hub_attention_probe.log
ROUND=r2 timeout 1300 bash tools/capture_profiles.sh > gpurun_out/capture.log 2>&1; tail -c 400 gpurun_out/capture.log
timeout 600 python tools/emulate_sharded.py --world 8 --single-gpu-ms 6.98 --link-gbps 61 76.8 > gpurun_out/emulate_r2d.log 2>&1; echo "emulate rc=$?"
timeout 300 python tools/emulate_sharded.py --world 4 --single-gpu-ms 6.98 --link-gbps 61 --out gpurun_out/emulated_sharded_w4.json > gpurun_out/emulate_r2d_w4.log 2>&1; echo "emulate4 rc=$?"
