cd /root/repo
O=/root/repo/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $O/pmc_bzw
AHIP_BZ_WALK_WGS=28 timeout -k 5 300 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --output-format csv -d $O/pmc_bzw -o w -- python /root/repo/tests/perf/bzip2_stats.py 384 > $O/pmc_bzw.log 2>&1
python - <<'PY'
import csv, glob, collections
f = glob.glob('/root/repo/gpurun_out/pmc_bzw/**/*counter_collection.csv', recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f[0])):
    k = r['Kernel_Name'][:40]
    acc[k][r['Counter_Name']] += float(r['Counter_Value'])
seen=set()
for r in csv.DictReader(open(f[0])):
    key=(r['Kernel_Name'][:40], r['Dispatch_Id'])
    if key not in seen: seen.add(key); n[key[0]] += 1
for k, d in acc.items():
    if 'bz_' in k: print(k, n[k], {c: "%.3g" % (v / n[k]) for c, v in d.items()})
PY
