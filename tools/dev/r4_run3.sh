# Round 4, GPU call 3: prefetched staging + retire batches of 8, shards call on worker streams, bench with the strong headline.
set -x
cd /root/repo
O=/root/repo/gpurun_out
mkdir -p $O
timeout -k 5 900 python -m pytest tests -m gpu -q -x 2>&1 | grep -v "RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl path" | tail -8 | tee $O/r4_pytest3.log
timeout -k 5 300 python bench.py --steps 10 --warmup 2 --cpu-seconds 0 --no-extras > $O/r4_bench3.log 2>&1; tail -1 $O/r4_bench3.log | cut -c1-300
( AHIP_KTIME=1 timeout 200 python tools/kstats.py 65536 log 2>&1 | grep -v amdgpu.ids | tail -8
for v in u64 u256e16 e4 e16; do echo "=== variant $v"; AHIP_KTIME=1 AHIP_LIB=/root/repo/archive_amd/lib/var_$v.so timeout 200 python tools/kstats.py 65536 log 2>&1 | grep "ktime\|kernel " | tail -2; done ) > $O/r4_occ3.log 2>&1
grep -v "^+" $O/r4_occ3.log | tail -30
