#!/bin/bash
# Dev tool (GPU box): SQ counters of the resolver for library variants (tools/mkvariants.py):  tools/wg_sq.sh <tag> name...
# One --pmc pass per variant (kernel-trace only next to --pmc); summary by tools/sq_summary.py <tag>_<name>.
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in "$@"; do
  rm -rf $R/gpurun_out/sq_${tag}_${v}_1 $R/gpurun_out/sq_${tag}_${v}_2
  AHIP_LIB=$R/archive_amd/lib/var_$v.so timeout -k 5 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY --output-format csv -d $R/gpurun_out/sq_${tag}_${v}_1 -o sq -- python $R/tools/ablate.py 65536 > $R/gpurun_out/sq_${tag}_${v}_1.log 2>&1
  (cd $R && python tools/sq_summary.py ${tag}_${v} resolve | grep -A12 resolve)
done
