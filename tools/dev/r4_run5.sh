# Round 4, GPU call 5: deterministic Deflate (sizes, speed), kernel split of one long member
set -x
cd /root/repo
O=/root/repo/gpurun_out
mkdir -p $O
timeout -k 5 1200 python -m pytest tests -m gpu -q -x 2>&1 | grep -v "RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl path" | tail -15 | tee $O/r4_pytest5.log
( timeout 300 python tools/deflate_sizes.py 2>&1 | grep -v amdgpu.ids | tail -20
timeout 300 python tests/perf/deflate_stats.py 1024 2>&1 | grep -v amdgpu.ids | tail -8 ) > $O/r4_df5.log 2>&1
cat $O/r4_df5.log
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof_sm4
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_sm4 -o sm -- python /root/repo/tools/sm_check.py 256 wiki > $O/prof_sm4.log 2>&1
python - <<'PY'
import csv, glob
f = glob.glob('/root/repo/gpurun_out/prof_sm4/**/*kernel_stats.csv', recursive=True)
if f:
    rows = list(csv.DictReader(open(f[0])))
    for r in rows[:16]:
        print("%-60s calls %5s avg %10.1f us max %10.1f us" % (r['Name'][:60], r['Calls'], float(r['AverageNs'])/1e3, float(r['MaxNs'])/1e3))
PY
tail -6 $O/prof_sm4.log
