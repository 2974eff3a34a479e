cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gls_gpu.py tests/test_gls_fuzz_gpu.py -m gpu -q -x 2>&1 | tail -2
for rep in 1 2; do
for v in _prev _new; do
  export GNNGLS_HIP_SO=$PWD/gnngls_amd/libgnngls_hip$v.so
  echo "=== variant '$v'"
  timeout 120 python scripts/probe_gls.py 200 256 2.0 0 model 2>&1 | grep "n="
  timeout 120 python scripts/probe_gls.py 150 256 2.0 0 model 2>&1 | grep "n="
done
done
