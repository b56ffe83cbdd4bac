# Refresh of what changed late in a round without touching the inflate kernels (run on the GPU box through gpurun):
# the GPU parity tests, the bench line with its extras, and the side profiles of Deflate and of one long member.
#   bash tools/run_final_prof.sh r05      then, here:  python -c "from tools import prof_summary as p; p.side_only('r05')"
R=${1:-r05}
O=/root/repo/gpurun_out
mkdir -p $O
cd /root/repo
timeout -k 5 900 python -m pytest tests -m gpu -q 2>&1 | grep -v "RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl path" | tail -2 | tee $O/pytest_gpu_$R.log
timeout -k 5 400 python bench.py --steps 10 --warmup 2 --cpu-seconds 12 > $O/bench_$R.log 2>&1; tail -1 $O/bench_$R.log | cut -c1-600
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof_df $O/prof_sm
timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_df -o df -- python /root/repo/tests/perf/deflate_stats.py 1024 > $O/prof_df.log 2>&1
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_sm -o sm -- python /root/repo/tools/sm_check.py 256 wiki > $O/prof_sm.log 2>&1
find $O/prof_df $O/prof_sm -name "*kernel_trace*" -delete
grep "deflate L6" $O/prof_df.log | tail -1; grep "gzip_decode_device" $O/prof_sm.log | tail -1
