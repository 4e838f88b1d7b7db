mkdir -p gpurun_out
PYGSD_SPMM_PACKED=1 timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -4
timeout 600 python tools/packed_probe.py > gpurun_out/packed_probe.log 2>&1; tail -50 gpurun_out/packed_probe.log
