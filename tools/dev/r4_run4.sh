# Round 4, GPU call 4: long members with history in front of them (q8), stored-block starts in the block finder, shards on worker streams.
set -x
cd /root/repo
O=/root/repo/gpurun_out
mkdir -p $O
timeout -k 5 1200 python -m pytest tests -m gpu -q -x 2>&1 | grep -v "RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl path" | tail -15 | tee $O/r4_pytest4.log
( AHIP_KTIME=1 timeout 200 python tools/kstats.py 65536 log 2>&1 | grep "ktime\|kernel " | tail -3
timeout 300 python tools/sm_check.py 64 noise 2>&1 | grep -v amdgpu.ids | tail -12
timeout 300 python tools/sm_check.py 256 wiki 2>&1 | grep "gzip_decode_device" ) > $O/r4_occ4.log 2>&1
grep -v "^+" $O/r4_occ4.log | tail -30
