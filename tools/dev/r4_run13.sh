set -x
cd /root/repo
O=/root/repo/gpurun_out
timeout -k 5 600 python -m pytest tests/test_single_stream_gpu.py tests/test_inflate_gpu.py tests/test_gzip_gpu.py -m gpu -q -x 2>&1 | grep -v "RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl path" | tail -4 | tee $O/r4_pytest13.log
for S in 4 8 2; do
echo "== split $S"
AHIP_SM_SPLIT=$S timeout 300 python tools/sm_check.py 256 wiki 2>&1 | grep "gzip_decode_device\|device bytes"
done
echo "== split 8 chunk 64K"; AHIP_SM_CHUNK=65536 AHIP_SM_SPLIT=8 timeout 300 python tools/sm_check.py 256 wiki 2>&1 | grep "gzip_decode_device\|device bytes"
echo "== split 8 chunk 32K"; AHIP_SM_CHUNK=32768 AHIP_SM_SPLIT=8 timeout 300 python tools/sm_check.py 256 wiki 2>&1 | grep "gzip_decode_device\|device bytes"
timeout -k 5 300 python bench.py --steps 10 --warmup 2 --cpu-seconds 0 --no-extras > $O/r4_bench13.log 2>&1; tail -1 $O/r4_bench13.log | cut -c1-200
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof_idx
timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_idx -o idx -- python /root/repo/bench.py --steps 5 --warmup 1 --cpu-seconds 0 --no-extras > $O/prof_idx.log 2>&1
python - <<'PY'
import csv, glob
f = glob.glob('/root/repo/gpurun_out/prof_idx/**/*kernel_stats.csv', recursive=True)
rows = list(csv.DictReader(open(f[0])))
for r in rows[:8]:
    print("%-50s calls %4s avg %9.1f us" % (r['Name'][:50], r['Calls'], float(r['AverageNs'])/1e3))
PY
rm -rf $O/prof_sm13
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_sm13 -o sm -- python /root/repo/tools/sm_check.py 256 wiki > $O/prof_sm13.log 2>&1
python - <<'PY'
import csv, glob
f = glob.glob('/root/repo/gpurun_out/prof_sm13/**/*kernel_stats.csv', recursive=True)
rows = list(csv.DictReader(open(f[0])))
for r in rows:
    if 'sm_' in r['Name']: print("%-50s calls %4s avg %9.1f us" % (r['Name'][:50], r['Calls'], float(r['AverageNs'])/1e3))
PY
