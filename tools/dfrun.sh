python tools/deflate_quick.py 2>&1 | grep level; python tools/deflate_sizes.py 2>&1 | grep -v amdgpu
timeout 600 python -m pytest tests/test_deflate_gpu.py tests/test_full_size_gpu.py -m gpu -q 2>&1 | grep -v "RCCL\|HIP version\|ROCm\|Hostname\|Librccl" | tail -6
