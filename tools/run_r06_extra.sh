# Round 6 side passes (GPU box, through gpurun): the workgroup-per-member resolver (AHIP_RES_WG=1) under the same kernel-trace
# and traffic passes as the production kernels, and the memory-side traffic of a bzip2 decode (55 MiB = 64 blocks of 900 k).
#   bash tools/run_r06_extra.sh
set -x
cd /root/repo
O=/root/repo/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python /root/repo/bench.py --cpu-seconds 0 --no-extras"
rm -rf $O/wg_prof $O/wg_rq $O/wg_wq $O/bz_rq $O/bz_wq $O/bz_prof64
AHIP_RES_WG=1 timeout -k 5 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/wg_prof -o wg -- $B --steps 3 --warmup 1 > $O/wg_prof.log 2>&1
AHIP_RES_WG=1 timeout -k 5 120 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --output-format csv -d $O/wg_rq -o q -- $B --steps 2 --warmup 1 > $O/wg_rq.log 2>&1
AHIP_RES_WG=1 timeout -k 5 120 rocprofv3 --kernel-trace --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --output-format csv -d $O/wg_wq -o q -- $B --steps 2 --warmup 1 > $O/wg_wq.log 2>&1
timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bz_prof64 -o bz -- python /root/repo/tests/perf/bzip2_stats.py 55 > $O/bz_prof64.log 2>&1
timeout -k 5 200 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --output-format csv -d $O/bz_rq -o q -- python /root/repo/tests/perf/bzip2_stats.py 55 > $O/bz_rq.log 2>&1
timeout -k 5 200 rocprofv3 --kernel-trace --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --output-format csv -d $O/bz_wq -o q -- python /root/repo/tests/perf/bzip2_stats.py 55 > $O/bz_wq.log 2>&1
ls $O | head -60
