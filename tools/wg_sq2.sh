#!/bin/bash
# Dev tool (GPU box): a second set of SQ counters (latency levels, residency, LDS stalls) of the resolver:  tools/wg_sq2.sh <tag> name...
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in "$@"; do
  rm -rf $R/gpurun_out/sq_${tag}b_${v}_1 $R/gpurun_out/sq_${tag}b_${v}_2
  AHIP_LIB=$R/archive_amd/lib/var_$v.so timeout -k 5 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_LEVEL_WAVES SQ_BUSY_CU_CYCLES SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_IFETCH SQ_INSTS_VMEM_RD SQ_INSTS_LDS --output-format csv -d $R/gpurun_out/sq_${tag}b_${v}_1 -o sq -- python $R/tools/ablate.py 65536 > $R/gpurun_out/sq_${tag}b_${v}_1.log 2>&1
  AHIP_LIB=$R/archive_amd/lib/var_$v.so timeout -k 5 300 rocprofv3 --kernel-trace --pmc SQ_LDS_UNALIGNED_STALL SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH SQ_INSTS_SMEM --output-format csv -d $R/gpurun_out/sq_${tag}b_${v}_2 -o sq -- python $R/tools/ablate.py 65536 > $R/gpurun_out/sq_${tag}b_${v}_2.log 2>&1
  (cd $R && python tools/sq_summary.py ${tag}b_${v} resolve | grep -A18 resolve_wg)
done
