cd /root/repo
O=/root/repo/gpurun_out
R=r04
timeout -k 5 900 python -m pytest tests -m gpu -q 2>&1 | grep -v "RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl path" | tail -2 | tee $O/pytest_gpu_$R.log
timeout -k 5 500 python bench.py --steps 10 --warmup 2 --cpu-seconds 12 > $O/bench_$R.log 2>&1; tail -1 $O/bench_$R.log | cut -c1-200
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof_df
timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_df -o df -- python /root/repo/tests/perf/deflate_stats.py 1024 > $O/prof_df.log 2>&1
tail -3 $O/prof_df.log
