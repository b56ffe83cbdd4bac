# SQ counter passes for the two inflate kernels (run on the GPU box through gpurun).
#   bash tools/run_sq.sh r02a
# Each pass is its own rocprofv3 run with --kernel-trace only (no other trace domains next to --pmc).
set -x
R=${1:-r02}
cd /tmp && export TMPDIR=/tmp
O=/root/repo/gpurun_out
mkdir -p $O
rocprofv3 -L > $O/counters_list.txt 2>&1
CMD="python /root/repo/bench.py --steps 2 --warmup 1 --cpu-seconds 0"
i=0
for SET in \
  "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
  "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_SMEM" \
  "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_FLAT SQ_INSTS_BRANCH SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC" \
  "GRBM_GUI_ACTIVE GRBM_COUNT TCC_HIT_sum TCC_MISS_sum"
do
  i=$((i+1))
  rm -rf $O/sq_${R}_$i
  timeout 240 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $O/sq_${R}_$i -o sq -- $CMD > $O/sq_${R}_$i.log 2>&1
  tail -2 $O/sq_${R}_$i.log
done
find $O -name "*counter_collection.csv" | head
