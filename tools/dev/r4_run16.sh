cd /root/repo
for C in 4 7 14 20; do
echo "== walk wgs per xcd $C"
AHIP_BZ_WALK_WGS=$C timeout 300 python tests/perf/bzip2_stats.py 384 2>&1 | grep "device-resident" | tail -1
done
