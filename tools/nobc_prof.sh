cd /tmp && export TMPDIR=/tmp
rm -rf /root/repo/gpurun_out/prof_nobc4
timeout -k 5 150 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d /root/repo/gpurun_out/prof_nobc4 -o nb -- python /root/repo/bench.py --no-bc --no-extras --cpu-seconds 0 --steps 5 --warmup 1 > /root/repo/gpurun_out/prof_nobc4.log 2>&1
tail -1 /root/repo/gpurun_out/prof_nobc4.log | cut -c1-200
python - <<PY
import csv,glob
for f in glob.glob("/root/repo/gpurun_out/prof_nobc4/**/*_stats.csv",recursive=True):
    print(f.split("/")[-1])
    for r in list(csv.DictReader(open(f)))[:14]: print("  ", r["Name"][:60], r["Calls"], "avg ms %.3f"%(float(r["AverageNs"])/1e6), "total ms %.2f"%(float(r["TotalDurationNs"])/1e6))
PY
