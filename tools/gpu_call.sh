timeout 600 python -m pytest tests/test_gpu_layers.py tests/test_gpu_kernels.py -m gpu -q -p no:cacheprovider 2>&1 | tail -4
timeout 300 python tools/hub_build_probe.py 2>&1 | tail -3
