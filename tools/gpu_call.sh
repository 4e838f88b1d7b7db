mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -p no:cacheprovider -x -k "dense or rows_per_wavefront" 2>&1 | tail -4
timeout 600 python tools/dense_probe.py 2>&1 | tail -6
