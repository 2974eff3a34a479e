cd $GRAFT_REPO_ROOT
echo "--- overflow-path build (GLS_QUIET_PEND_CAP=2)"
GNNGLS_HIP_SO=$PWD/gnngls_amd/libgnngls_hip_pendcap2.so timeout 1200 python -m pytest tests/test_gls_gpu.py tests/test_gls_fuzz_gpu.py -m gpu -q -x 2>&1 | tail -2
GNNGLS_HIP_SO=$PWD/gnngls_amd/libgnngls_hip_pendcap2.so timeout 120 python scripts/probe_gls.py 100 1024 2.0 0 model 2>&1 | grep "n="
echo "--- product build"
timeout 120 python scripts/probe_gls.py 100 1024 2.0 0 model 2>&1 | grep "n="
python __graft_entry__.py smoke 2>&1 | tail -1
timeout 900 python -m pytest tests/test_perf_floor_gpu.py tests/test_bench_gpu.py -m gpu -q 2>&1 | tail -3
