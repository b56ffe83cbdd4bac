cd /root/repo
O=/root/repo/gpurun_out
cd /tmp && export TMPDIR=/tmp
for v in gzh_noslow gzh_nostore; do
rm -rf $O/prof_$v
AHIP_LIB=/root/repo/archive_amd/lib/var_$v.so timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$v -o x -- python /root/repo/bench.py --steps 3 --warmup 1 --cpu-seconds 0 --no-extras > $O/prof_$v.log 2>&1
echo "== $v"; grep "gz_parse_headers\|gz_count" $O/prof_$v/*/x_kernel_stats.csv $O/prof_$v/x_kernel_stats.csv 2>/dev/null | cut -d, -f1-4 | head -3
done
