cd /tmp && export TMPDIR=/tmp
O=/root/repo/gpurun_out
rm -rf $O/sq_nobc_1
timeout -k 5 120 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU GRBM_GUI_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD --output-format csv -d $O/sq_nobc_1 -o sq -- python /root/repo/bench.py --no-bc --no-extras --cpu-seconds 0 --steps 2 --warmup 1 > $O/sq_nobc_1.log 2>&1
cd /root/repo; python tools/sq_summary.py nobc | grep -v "late_kernel" | head -40
