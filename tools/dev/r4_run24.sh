cd /root/repo
O=/root/repo/gpurun_out
timeout -k 5 1500 python -m pytest tests -m gpu -q 2>&1 | grep -v "RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl path" | tail -8 | tee $O/r4_pytest24.log
