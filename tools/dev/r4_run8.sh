set -x
cd /root/repo
O=/root/repo/gpurun_out
timeout -k 5 1200 python -m pytest tests -m gpu -q -x 2>&1 | grep -v "RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl path" | tail -30 | tee $O/r4_pytest8.log
