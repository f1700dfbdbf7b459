set -u
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "dense_split" 2>&1 | tail -3
python tools/run_dense_once.py 2>&1 | grep -v amdgpu.ids
