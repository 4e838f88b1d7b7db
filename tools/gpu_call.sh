mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_sharded.py -m gpu -q -p no:cacheprovider 2>&1 | tail -2
timeout 900 python tools/emulate_sharded_c5.py 2>&1 | grep -v Warn | cut -c1-330 | tail -3
