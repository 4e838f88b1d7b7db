mkdir -p gpurun_out
timeout 600 python tools/emulate_sharded.py --world 8 --single-gpu-ms 6.98 --link-gbps 61 76.8 > gpurun_out/emulate_r2e.log 2>&1; echo "emulate rc=$?"
timeout 300 python tools/emulate_sharded.py --world 4 --single-gpu-ms 6.98 --link-gbps 61 --out gpurun_out/emulated_sharded_w4.json > gpurun_out/emulate_r2e_w4.log 2>&1; echo "emulate4 rc=$?"
timeout 300 python tools/emulate_sharded.py --world 8 --single-gpu-ms 6.98 --link-gbps 61 --delay spin --out gpurun_out/emulated_sharded_spin.json > gpurun_out/emulate_r2e_spin.log 2>&1; echo "emulate-spin rc=$?"
timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "id_range" -p no:cacheprovider 2>&1 | tail -3
ROUND=r2 timeout 1300 bash tools/capture_profiles.sh > gpurun_out/capture.log 2>&1; tail -c 300 gpurun_out/capture.log
