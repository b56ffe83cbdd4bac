# Round 4, GPU call 1: parity of the new member-index kernels + residency sweeps of the two inflate kernels.
set -x
cd /root/repo
O=/root/repo/gpurun_out
mkdir -p $O
timeout -k 5 900 python -m pytest tests -m gpu -q -x 2>&1 | grep -v "RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl path" | tail -5 | tee $O/r4_pytest1.log
timeout -k 5 300 python bench.py --steps 10 --warmup 2 --cpu-seconds 0 --no-extras > $O/r4_bench1.log 2>&1; tail -1 $O/r4_bench1.log | cut -c1-300
( AHIP_KTIME=1 timeout 200 python tools/kstats.py 65536 log 2>&1 | grep -v amdgpu.ids | head -20
for w in 6 8 9; do echo "=== tok wgs/cu $w"; AHIP_KTIME=1 AHIP_TOK_WGS_PER_CU=$w timeout 200 python tools/kstats.py 65536 log 2>&1 | grep "ktime\|kernel " | tail -3; done
for w in 10 14 17; do echo "=== res wgs/cu $w"; AHIP_KTIME=1 AHIP_RES_WGS_PER_CU=$w timeout 200 python tools/kstats.py 65536 log 2>&1 | grep "ktime\|kernel " | tail -3; done ) > $O/r4_occ1.log 2>&1
cat $O/r4_occ1.log | tail -60
