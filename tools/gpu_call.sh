timeout 900 python -m pytest tests/test_gpu_sharded.py -m gpu -q -p no:cacheprovider 2>&1 | tail -2
