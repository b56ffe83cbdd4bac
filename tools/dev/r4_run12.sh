set -x
cd /root/repo
O=/root/repo/gpurun_out
export AHIP_KTIME=1
bash tools/exp.sh 65536 log w16 w12 w12r96 2>&1 | grep -v "^per member\|status hist\|debug codes" | tee $O/r4_occ12.log
unset AHIP_KTIME
timeout -k 5 300 python bench.py --steps 10 --warmup 2 --cpu-seconds 0 --no-extras > $O/r4_bench12.log 2>&1; tail -1 $O/r4_bench12.log | cut -c1-200
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof_idx
timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_idx -o idx -- python /root/repo/bench.py --steps 5 --warmup 1 --cpu-seconds 0 --no-extras > $O/prof_idx.log 2>&1
python - <<'PY'
import csv, glob
f = glob.glob('/root/repo/gpurun_out/prof_idx/**/*kernel_stats.csv', recursive=True)
rows = list(csv.DictReader(open(f[0])))
for r in rows[:14]:
    print("%-50s calls %4s avg %9.1f us" % (r['Name'][:50], r['Calls'], float(r['AverageNs'])/1e3))
PY
