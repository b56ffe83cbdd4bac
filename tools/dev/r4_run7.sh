set -x
cd /root/repo
O=/root/repo/gpurun_out
timeout -k 5 1200 python -m pytest tests -m gpu -q -x 2>&1 | grep -v "RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl path" | tail -6 | tee $O/r4_pytest7.log
timeout -k 5 300 python bench.py --steps 10 --warmup 2 --cpu-seconds 0 --no-extras > $O/r4_bench7.log 2>&1; tail -1 $O/r4_bench7.log | cut -c1-260
( AHIP_KTIME=1 timeout 200 python tools/kstats.py 65536 log 2>&1 | grep "ktime\|kernel " | tail -3
for w in 11 12 13; do echo "=== tok wgs/cu $w"; AHIP_KTIME=1 AHIP_TOK_WGS_PER_CU=$w timeout 200 python tools/kstats.py 65536 log 2>&1 | grep "ktime\|kernel " | tail -2; done
echo "=== wiki"; AHIP_KTIME=1 timeout 200 python tools/kstats.py 16384 wiki 2>&1 | grep "ktime\|kernel " | tail -2 ) > $O/r4_occ7.log 2>&1
grep -v "^+" $O/r4_occ7.log | tail -20
