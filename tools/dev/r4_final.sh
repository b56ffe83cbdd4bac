cd /root/repo
O=/root/repo/gpurun_out
R=r04
timeout -k 5 900 python -m pytest tests -m gpu -q 2>&1 | grep -v "RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl path" | tail -2 | tee $O/pytest_gpu_$R.log
timeout -k 5 500 python bench.py --steps 10 --warmup 2 --cpu-seconds 12 > $O/bench_$R.log 2>&1; tail -1 $O/bench_$R.log | cut -c1-300
