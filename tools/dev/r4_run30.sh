cd /root/repo
for C in 28 40 56 80 112 160; do
echo "== walk wgs per xcd $C (64 blocks)"
AHIP_BZ_WALK_WGS=$C timeout 120 python tests/perf/bzip2_stats.py 55 2>&1 | grep "device-resident" | tail -1
done
