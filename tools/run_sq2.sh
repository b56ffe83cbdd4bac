# SQ counter passes over tools/kstats.py (one decode set) for a given library variant
#   bash tools/run_sq2.sh <variant> <tag> [members]
V=$1; R=$2; M=${3:-16384}
cd /tmp && export TMPDIR=/tmp
O=/root/repo/gpurun_out
export AHIP_LIB=/root/repo/archive_amd/lib/var_$V.so
python /root/repo/tools/kstats.py $M log > /dev/null 2>&1   # warm the corpus cache
i=0
for SET in \
  "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
  "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_SMEM" \
  "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_BRANCH SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE"
do
  i=$((i+1))
  rm -rf $O/sq_${R}_$i
  timeout 120 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $O/sq_${R}_$i -o sq -- python /root/repo/tools/kstats.py $M log > $O/sq_${R}_$i.log 2>&1
done
