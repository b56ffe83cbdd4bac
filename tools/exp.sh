# Dev tool (GPU box): run kstats over every tuning variant archive_amd/lib/var_*.so (or the names given).
#   bash tools/exp.sh [members] [kind] [name ...]
M=${1:-65536}; K=${2:-log}; shift; shift
cd /root/repo
LIBS="$@"
if [ -z "$LIBS" ]; then LIBS=$(ls archive_amd/lib/var_*.so | sed 's/.*var_\(.*\)\.so/\1/'); fi
for n in $LIBS; do
  echo "=== $n"
  AHIP_LIB=/root/repo/archive_amd/lib/var_$n.so timeout 120 python tools/kstats.py $M $K 2>&1 | head -24
done
