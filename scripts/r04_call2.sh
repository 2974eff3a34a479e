# round 4, GPU call 2: exactness of the descent experiments (all on), then same-box A/B of each
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/r04b; mkdir -p $out
export GNNGLS_HIP_SO=$PWD/gnngls_amd/libgnngls_hip_e127.so
timeout 1200 python -m pytest tests/test_gls_gpu.py tests/test_gls_fuzz_gpu.py tests/test_search_progress_gpu.py -m gpu -q -x 2>&1 | tail -3 > $out/pytest_e127.log
cat $out/pytest_e127.log
export GNNGLS_HIP_SO=$PWD/gnngls_amd/libgnngls_hip_e7.so
timeout 1200 python -m pytest tests/test_gls_gpu.py tests/test_gls_fuzz_gpu.py -m gpu -q -x 2>&1 | tail -3 > $out/pytest_e7.log
cat $out/pytest_e7.log
for rep in 1 2; do
for v in base e1 e2 e7 e127; do
  export GNNGLS_HIP_SO=$PWD/gnngls_amd/libgnngls_hip_$v.so
  echo "=== variant $v" >> $out/ab_descent.log
  timeout 120 python scripts/probe_gls.py 100 1024 2.0 0 model 2>&1 | grep "n=" >> $out/ab_descent.log
  timeout 120 python scripts/probe_gls.py 100 1024 2.0 0 noise 2>&1 | grep "n=" >> $out/ab_descent.log
  timeout 120 python scripts/probe_gls.py 200 256 2.0 0 model 2>&1 | grep "n=" >> $out/ab_descent.log
  timeout 120 python scripts/probe_gls.py 50 128 1.0 0 model 2>&1 | grep "n=" >> $out/ab_descent.log
  timeout 120 python scripts/probe_gls.py 50 2048 1.0 0 model 2>&1 | grep "n=" >> $out/ab_descent.log
done
done
cat $out/ab_descent.log
