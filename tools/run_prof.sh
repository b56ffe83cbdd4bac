# Round profile recipe (run on the GPU box through gpurun; every step under its own timeout).
#   bash tools/run_prof.sh r02
#   1. GPU parity tests, 2. the bench line (with extras), 3. rocprofv3 kernel-trace stats of the bench command,
#   4./5. FETCH_SIZE and WRITE_SIZE in separate --pmc passes (bench + a known-byte-count calibration),
#   6. SQ counters of the two inflate kernels in three --pmc passes (kernel-trace only next to --pmc),
#   7. side profiles: Deflate (config 3), one long member (config 2a), bzip2 (config 5), checksums.
# tools/prof_summary.py turns gpurun_out/ into the committed profiles/rNN_* summaries.
set -x
R=${1:-r06}
MODE=${2:-full}   # "core": only the passes over the bench command (3.-6.), each under a short timeout
T=200; [ "$MODE" = core ] && T=60
cd /root/repo
O=/root/repo/gpurun_out
mkdir -p $O
if [ "$MODE" != core ]; then
timeout -k 5 600 python -m pytest tests -m gpu -q 2>&1 | grep -v "RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl path" | tail -2 | tee $O/pytest_gpu_$R.log
timeout -k 5 400 python bench.py --steps 10 --warmup 2 --cpu-seconds 12 > $O/bench_$R.log 2>&1; tail -1 $O/bench_$R.log | cut -c1-400
fi
cd /tmp && export TMPDIR=/tmp
B="python /root/repo/bench.py --cpu-seconds 0 --no-extras"
rm -rf $O/prof_df $O/prof_bz $O/prof_sm $O/prof_$R $O/pmc_fetch $O/pmc_write $O/cal_fetch $O/cal_write $O/sq_${R}_*
timeout -k 5 $T rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$R -o $R -- $B --steps 5 --warmup 1 > $O/prof_$R.log 2>&1
timeout -k 5 $T rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o f -- $B --steps 3 --warmup 1 > $O/pmc_fetch.log 2>&1
timeout -k 5 $T rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o w -- $B --steps 3 --warmup 1 > $O/pmc_write.log 2>&1
timeout -k 5 $T rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/cal_fetch -o f -- python /root/repo/tools/pmc_calib.py > $O/cal_fetch.log 2>&1
timeout -k 5 $T rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/cal_write -o w -- python /root/repo/tools/pmc_calib.py > $O/cal_write.log 2>&1
# request-size split of the memory-side traffic (what FETCH_SIZE / WRITE_SIZE are derived from) + L2 hit rate, bench and calibration
rm -rf $O/pmc_rq $O/pmc_wq $O/cal_rq $O/cal_wq $O/pmc_l2
timeout -k 5 $T rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --output-format csv -d $O/pmc_rq -o q -- $B --steps 3 --warmup 1 > $O/pmc_rq.log 2>&1
timeout -k 5 $T rocprofv3 --kernel-trace --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --output-format csv -d $O/pmc_wq -o q -- $B --steps 3 --warmup 1 > $O/pmc_wq.log 2>&1
timeout -k 5 $T rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --output-format csv -d $O/cal_rq -o q -- python /root/repo/tools/pmc_calib.py > $O/cal_rq.log 2>&1
timeout -k 5 $T rocprofv3 --kernel-trace --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --output-format csv -d $O/cal_wq -o q -- python /root/repo/tools/pmc_calib.py > $O/cal_wq.log 2>&1
timeout -k 5 $T rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $O/pmc_l2 -o q -- $B --steps 3 --warmup 1 > $O/pmc_l2.log 2>&1
i=0
for SET in \
  "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
  "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_SMEM" \
  "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_BRANCH SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE"
do
  i=$((i+1))
  timeout -k 5 $T rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $O/sq_${R}_$i -o sq -- $B --steps 2 --warmup 1 > $O/sq_${R}_$i.log 2>&1
done
[ "$MODE" = core ] && { ls $O | head -40; exit 0; }
timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_df -o df -- python /root/repo/tests/perf/deflate_stats.py 1024 > $O/prof_df.log 2>&1
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_sm -o sm -- python /root/repo/tools/sm_check.py 256 wiki > $O/prof_sm.log 2>&1
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bz -o bz -- python /root/repo/tests/perf/bzip2_stats.py 384 > $O/prof_bz.log 2>&1
timeout -k 5 120 python /root/repo/tools/checksum_stats.py 1024 > $O/checksum_stats.log 2>&1; tail -2 $O/checksum_stats.log
ls $O | head -40
