# Round profile recipe (run on the GPU box through gpurun; every step under its own timeout).
#   1. GPU parity tests, 2. the bench line, 3. rocprofv3 kernel-trace stats of the bench command,
#   4./5. FETCH_SIZE and WRITE_SIZE in separate --pmc passes (bench + a known-byte-count calibration).
# tools/prof_summary.py turns gpurun_out/ into the committed profiles/rNN_* summaries.
set -x
R=${1:-r01}
cd /root/repo
timeout 300 python -m pytest tests -m gpu -q 2>&1 | tail -2
timeout 280 python bench.py --steps 10 --warmup 2 --cpu-seconds 12 > gpurun_out/bench_$R.log 2>&1; tail -1 gpurun_out/bench_$R.log
cd /tmp && export TMPDIR=/tmp
O=/root/repo/gpurun_out
rm -rf $O/prof_df $O/prof_bz $O/prof_sm $O/prof_$R $O/pmc_fetch $O/pmc_write $O/cal_fetch $O/cal_write
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$R -o $R -- python /root/repo/bench.py --steps 5 --warmup 1 --cpu-seconds 0 > $O/prof_$R.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o f -- python /root/repo/bench.py --steps 3 --warmup 1 --cpu-seconds 0 > $O/pmc_fetch.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o w -- python /root/repo/bench.py --steps 3 --warmup 1 --cpu-seconds 0 > $O/pmc_write.log 2>&1
timeout 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/cal_fetch -o f -- python /root/repo/tools/pmc_calib.py > $O/cal_fetch.log 2>&1
timeout 120 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/cal_write -o w -- python /root/repo/tools/pmc_calib.py > $O/cal_write.log 2>&1
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_df -o df -- python /root/repo/tests/perf/deflate_stats.py 1024 > $O/prof_df.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_sm -o sm -- python /root/repo/tools/sm_check.py 256 wiki > $O/prof_sm.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bz -o bz -- python /root/repo/tests/perf/bzip2_stats.py 384 > $O/prof_bz.log 2>&1
timeout 120 python /root/repo/tools/checksum_stats.py 1024 > $O/checksum_stats.log 2>&1; tail -2 $O/checksum_stats.log
find $O/prof_$R $O/pmc_fetch $O/pmc_write $O/cal_fetch $O/cal_write -type f | head -30
