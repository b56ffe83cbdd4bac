#!/bin/bash
# Dev tool (GPU box): kernel times of the headline decode for a list of library variants (tools/mkvariants.py), nothing checked.
#   tools/wg_abl.sh <out dir> name1 name2 ...
out=$1; shift
mkdir -p $out
for v in "$@"; do
  AHIP_LIB=archive_amd/lib/var_$v.so AHIP_KTIME=1 timeout 300 python tools/ablate.py 65536 2>&1 | grep ktime | tail -2 | sed "s/^/$v: /" >> $out/abl.log
done
cat $out/abl.log
