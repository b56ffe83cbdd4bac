cd /root/repo
O=/root/repo/gpurun_out
timeout 300 python tests/perf/bzip2_stats.py 55 2>&1 | grep "device-resident\|ok=\|bz2 -9" | tail -4
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof_bz64
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bz64 -o bz -- python /root/repo/tests/perf/bzip2_stats.py 55 > $O/prof_bz64.log 2>&1
python - <<'PY'
import csv, glob
f = glob.glob('/root/repo/gpurun_out/prof_bz64/**/*kernel_stats.csv', recursive=True)
rows = list(csv.DictReader(open(f[0])))
tot=0
for r in rows[:24]:
    print("%-50s calls %4s avg %9.1f us" % (r['Name'][:50], r['Calls'], float(r['AverageNs'])/1e3))
PY
