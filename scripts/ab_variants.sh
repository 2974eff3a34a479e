cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gls_gpu.py tests/test_gls_fuzz_gpu.py -m gpu -q -x 2>&1 | tail -3
for thr in 512 1024; do
timeout 120 python scripts/probe_gls.py 200 256 2.0 0 noise $thr 2>&1 | grep "n="
timeout 120 python scripts/probe_gls.py 200 256 2.0 0 weight $thr 2>&1 | grep "n="
timeout 120 python scripts/probe_gls.py 150 256 2.0 0 noise $thr 2>&1 | grep "n="
done
