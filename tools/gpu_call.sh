mkdir -p gpurun_out
timeout 600 python tools/emulate_sharded.py --world 8 --single-gpu-ms 6.96 --link-gbps 61 76.8 --out gpurun_out/emulated_sharded.json > gpurun_out/emulate_r2g.log 2>&1; echo rc=$?
timeout 300 python tools/emulate_sharded.py --world 4 --single-gpu-ms 6.96 --link-gbps 61 --out gpurun_out/emulated_sharded_w4.json > gpurun_out/emulate_r2g_w4.log 2>&1; echo rc=$?
