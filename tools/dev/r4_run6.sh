# Round 4, GPU call 6: where exactly the 12th resident tokenizer wave is lost (LDS per workgroup), single long member again
set -x
cd /root/repo
O=/root/repo/gpurun_out
( for v in r1280 r1280p256 r1280p512 r1280p768; do echo "=== variant $v"; AHIP_KTIME=1 AHIP_LIB=/root/repo/archive_amd/lib/var_$v.so timeout 200 python tools/kstats.py 65536 log 2>&1 | grep "ktime\|kernel " | tail -2; done
AHIP_DEBUG=1 timeout 300 python tools/sm_check.py 256 wiki 2>&1 | grep "sm:\|gzip_decode_device" | tail -14 ) > $O/r4_occ6.log 2>&1
grep -v "^+" $O/r4_occ6.log | tail -40
