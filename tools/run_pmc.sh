# PMC passes over tools/kstats.py for one library variant: bash tools/run_pmc.sh <variant> <tag> <members> "<counters>" ["<counters>" ...]
V=$1; R=$2; M=$3; shift; shift; shift
cd /tmp && export TMPDIR=/tmp
O=/root/repo/gpurun_out
export AHIP_LIB=/root/repo/archive_amd/lib/var_$V.so
python /root/repo/tools/kstats.py $M log > /dev/null 2>&1   # warm the corpus cache
i=0
for SET in "$@"; do
  i=$((i+1))
  rm -rf $O/sq_${R}_$i
  timeout 150 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $O/sq_${R}_$i -o sq -- python /root/repo/tools/kstats.py $M log > $O/sq_${R}_$i.log 2>&1
done
