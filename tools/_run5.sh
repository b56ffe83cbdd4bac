set -x
mkdir -p gpurun_out/r5f
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python tools/df_same_bytes.py archive_amd/lib/var_prev.so archive_amd/lib/libarchive_hip.so > gpurun_out/r5f/same.log 2>&1; tail -3 gpurun_out/r5f/same.log
for v in libarchive_hip var_prev; do AHIP_LIB=$PWD/archive_amd/lib/$v.so timeout 300 python tools/deflate_quick.py 2>&1 | grep level; done > gpurun_out/r5f/quick.log; cat gpurun_out/r5f/quick.log
timeout 300 python -m pytest tests/test_deflate_gpu.py -m gpu -x -q 2>&1 | tail -2
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r5f/df -- python tests/perf/deflate_stats.py 1024 > gpurun_out/r5f/df.log 2>&1; tail -5 gpurun_out/r5f/df.log
find gpurun_out/r5f -name "*kernel_trace*" -delete
