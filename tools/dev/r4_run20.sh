cd /root/repo
O=/root/repo/gpurun_out
timeout -k 5 900 python -m pytest tests/test_bzip2.py -m gpu -q -x 2>&1 | grep -v "RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl path" | tail -4 | tee $O/r4_pytest20.log
echo "== nt pre"; AHIP_LIB=/root/repo/archive_amd/lib/var_ntpre.so timeout 300 python tests/perf/bzip2_stats.py 384 2>&1 | grep "device-resident\|ok=" | tail -2
echo "== production"; timeout 300 python tests/perf/bzip2_stats.py 384 2>&1 | grep "device-resident\|ok=" | tail -2
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof_bz14
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bz14 -o bz -- python /root/repo/tests/perf/bzip2_stats.py 384 > $O/prof_bz14.log 2>&1
python - <<'PY'
import csv, glob
f = glob.glob('/root/repo/gpurun_out/prof_bz14/**/*kernel_stats.csv', recursive=True)
rows = list(csv.DictReader(open(f[0])))
for r in rows[:14]:
    print("%-50s calls %4s avg %9.1f us" % (r['Name'][:50], r['Calls'], float(r['AverageNs'])/1e3))
PY
