mkdir -p gpurun_out
rm -f gpurun_out/parity_errors_*.json
( timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=8 > gpurun_out/pytest_gpu_r2c.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu_r2c.log )
tail -22 gpurun_out/pytest_gpu_r2c.log
timeout 300 python tools/hub_attention_probe.py > gpurun_out/hub_attention_probe.log 2>&1; tail -5 gpurun_out/hub_attention_probe.log
timeout 400 python bench.py > gpurun_out/bench_r2c.json 2> gpurun_out/bench_r2c.err; echo "bench rc=$?"
timeout 600 python tools/emulate_sharded.py --world 8 --single-gpu-ms 6.98 --link-gbps 61 76.8 > gpurun_out/emulate_r2c.log 2>&1; echo "emulate rc=$?"
timeout 300 python tools/emulate_sharded.py --world 4 --single-gpu-ms 6.98 --link-gbps 61 --out gpurun_out/emulated_sharded_w4.json > gpurun_out/emulate_r2c_w4.log 2>&1; echo "emulate4 rc=$?"
timeout 900 python tools/bench_configs.py > gpurun_out/configs.log 2>&1; echo "configs rc=$?"; tail -3 gpurun_out/configs.log | cut -c1-300
