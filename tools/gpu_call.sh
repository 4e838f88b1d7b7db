timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -p no:cacheprovider 2>&1 | tail -3
