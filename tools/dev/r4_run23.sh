cd /root/repo
O=/root/repo/gpurun_out
timeout -k 5 900 python -m pytest tests/test_bzip2.py -m gpu -q -x 2>&1 | grep -v "RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl path" | tail -4 | tee $O/r4_pytest23.log
for S in 1 2 4 8; do
echo "== streams $S, 64 blocks"; AHIP_BZ_STREAMS=$S timeout 300 python tests/perf/bzip2_stats.py 55 2>&1 | grep "device-resident\|ok=" | tail -2
done
for S in 1 4 8; do
echo "== streams $S, 448 blocks"; AHIP_BZ_STREAMS=$S timeout 300 python tests/perf/bzip2_stats.py 384 2>&1 | grep "device-resident\|ok=" | tail -2
done
