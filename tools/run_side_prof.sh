# Side profiles only (run on the GPU box through gpurun): bzip2 (config 5: 448 and 64 blocks), one long member (config 2a),
# config 4 without the BC subfield.  tools/prof_summary.py turns gpurun_out/ into the committed profiles/rNN_* summaries.
#   bash tools/run_side_prof.sh r03
R=${1:-r03}
O=/root/repo/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof_bz $O/prof_bz64 $O/prof_sm $O/prof_nobc
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bz -o bz -- python /root/repo/tests/perf/bzip2_stats.py 384 > $O/prof_bz.log 2>&1
timeout -k 5 150 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bz64 -o bz -- python /root/repo/tests/perf/bzip2_stats.py 55 > $O/prof_bz64.log 2>&1
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_sm -o sm -- python /root/repo/tools/sm_check.py 256 wiki > $O/prof_sm.log 2>&1
timeout -k 5 150 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_nobc -o nb -- python /root/repo/bench.py --no-bc --no-extras --cpu-seconds 0 --steps 5 --warmup 1 > $O/prof_nobc.log 2>&1
grep "device-resident" $O/prof_bz.log $O/prof_bz64.log | tail -4; tail -1 $O/prof_nobc.log | cut -c1-200; grep "one member" $O/prof_sm.log | tail -2
