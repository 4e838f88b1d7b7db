timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -p no:cacheprovider -k "bf16" 2>&1 | tail -2
