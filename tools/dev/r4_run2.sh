# Round 4, GPU call 2: the 6 KiB ring (12 resident tokenizer waves per CU?) -- parity, then timing of it and its variants.
set -x
cd /root/repo
O=/root/repo/gpurun_out
mkdir -p $O
timeout -k 5 900 python -m pytest tests -m gpu -q -x 2>&1 | grep -v "RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl path" | tail -5 | tee $O/r4_pytest2.log
timeout -k 5 300 python bench.py --steps 10 --warmup 2 --cpu-seconds 0 --no-extras > $O/r4_bench2.log 2>&1; tail -1 $O/r4_bench2.log | cut -c1-300
( AHIP_KTIME=1 timeout 200 python tools/kstats.py 65536 log 2>&1 | grep -v amdgpu.ids | tail -12
for w in 10 11 12 13; do echo "=== tok wgs/cu $w"; AHIP_KTIME=1 AHIP_TOK_WGS_PER_CU=$w timeout 200 python tools/kstats.py 65536 log 2>&1 | grep "ktime\|kernel " | tail -2; done
for v in emit16 spec192 steps8 steps16 ring2048 ring1280; do echo "=== variant $v"; AHIP_KTIME=1 AHIP_LIB=/root/repo/archive_amd/lib/var_$v.so timeout 200 python tools/kstats.py 65536 log 2>&1 | grep "ktime\|kernel " | tail -2; done
echo "=== wiki"; AHIP_KTIME=1 timeout 200 python tools/kstats.py 16384 wiki 2>&1 | grep "ktime\|kernel " | tail -2 ) > $O/r4_occ2.log 2>&1
cat $O/r4_occ2.log | grep -v "^+" | tail -60
