mkdir -p gpurun_out
timeout 900 python tools/emulate_sharded_c5.py 2>&1 | grep -v Warn | cut -c1-420 | tail -4
