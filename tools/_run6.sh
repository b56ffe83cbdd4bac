set -x
mkdir -p gpurun_out/r5g
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python tools/df_same_bytes.py archive_amd/lib/var_prev.so archive_amd/lib/libarchive_hip.so > gpurun_out/r5g/same.log 2>&1; tail -3 gpurun_out/r5g/same.log
timeout 300 python tools/deflate_quick.py 2>&1 | grep level > gpurun_out/r5g/quick.log; cat gpurun_out/r5g/quick.log
timeout 300 python -m pytest tests/test_deflate_gpu.py -m gpu -x -q 2>&1 | tail -2
AHIP_DEBUG=1 AHIP_LIB=$PWD/archive_amd/lib/libarchive_hip_prof.so timeout 300 python tools/deflate_quick.py > gpurun_out/r5g/df_prof.log 2>&1; grep "encode kernel" gpurun_out/r5g/df_prof.log | sed -n 4,4p
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r5g/df -- python tests/perf/deflate_stats.py 1024 > gpurun_out/r5g/df.log 2>&1; grep "deflate L6" gpurun_out/r5g/df.log
find gpurun_out/r5g -name "*kernel_trace*" -delete
grep deflate gpurun_out/r5g/df/*/*kernel_stats.csv | cut -d, -f1-7 | cut -c1-60,100-
