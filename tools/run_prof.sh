set -x
cd /root/repo
timeout 300 python -m pytest tests -m gpu -q 2>&1 | tail -2
timeout 280 python bench.py --steps 10 --warmup 2 --cpu-seconds 12 > gpurun_out/bench_r01.log 2>&1; tail -1 gpurun_out/bench_r01.log
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/prof_r01 -o r01 -- python /root/repo/bench.py --steps 5 --warmup 1 --cpu-seconds 0 > /root/repo/gpurun_out/prof_r01.log 2>&1
ls -R /root/repo/gpurun_out/prof_r01 | head
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /root/repo/gpurun_out/pmc_fetch -o f -- python /root/repo/bench.py --steps 3 --warmup 1 --cpu-seconds 0 > /root/repo/gpurun_out/pmc_fetch.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /root/repo/gpurun_out/pmc_write -o w -- python /root/repo/bench.py --steps 3 --warmup 1 --cpu-seconds 0 > /root/repo/gpurun_out/pmc_write.log 2>&1
ls /root/repo/gpurun_out/pmc_fetch /root/repo/gpurun_out/pmc_write
