cd /root/repo
O=/root/repo/gpurun_out
timeout -k 5 900 python -m pytest tests/test_deflate_gpu.py -m gpu -q -x 2>&1 | grep -v "RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl path" | tail -6 | tee $O/r4_pytest29.log
timeout 300 python tests/perf/deflate_stats.py 1024 2>&1 | tail -12
