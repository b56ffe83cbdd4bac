cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_single_stream_gpu.py -m gpu -x -q 2>&1 | tail -2
AHIP_DEBUG=1 AHIP_LIB=$PWD/archive_amd/lib/libarchive_hip_prof.so timeout 200 python tools/sm_time.py 256 wiki 2>&1 | grep "sm find" | tail -1
python tools/sm_time.py 256 wiki
python tools/sm_time.py 256 log
