cd /root/repo
timeout -k 5 300 python -m pytest tests/test_single_stream_gpu.py tests/test_full_size_gpu.py tests/test_inflate_gpu.py -m gpu -x -q 2>&1 | grep -v "RCCL\|HIP version\|ROCm\|Hostname\|Librccl" | tail -4
cd /tmp && export TMPDIR=/tmp
AHIP_DEBUG=1 timeout -k 5 100 python /root/repo/tools/sm_dev.py 256 wiki 2>&1 | grep "gzip_decode\|one pass\|again\|bytes ok" | tail -4
rm -rf /root/repo/gpurun_out/prof_sm3
timeout -k 5 120 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/prof_sm3 -o sm -- python /root/repo/tools/sm_dev.py 256 wiki > /dev/null 2>&1
python - <<PY
import csv,glob
f=glob.glob("/root/repo/gpurun_out/prof_sm3/**/*kernel_stats.csv",recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:8]: print(r["Name"][:50], r["Calls"], "avg ms %.3f"%(float(r["AverageNs"])/1e6), r["Percentage"])
PY
cd /root/repo; bash tools/exp.sh 65536 log base 2>&1 | grep "kernel"
