cd $GRAFT_REPO_ROOT
out=gpurun_out/r04j; mkdir -p $out
timeout 1200 python -m pytest tests/test_gls_gpu.py tests/test_gls_fuzz_gpu.py tests/test_search_progress_gpu.py -m gpu -q -x 2>&1 | tail -3 > $out/pytest_buf.log
cat $out/pytest_buf.log
for rep in 1 2; do
for v in _nobuf ""; do
  export GNNGLS_HIP_SO=$PWD/gnngls_amd/libgnngls_hip$v.so
  echo "=== variant '$v'" >> $out/ab_penbuf.log
  timeout 120 python scripts/probe_gls.py 100 1024 2.0 0 model 2>&1 | grep "n=" >> $out/ab_penbuf.log
  timeout 120 python scripts/probe_gls.py 100 1024 2.0 0 noise 2>&1 | grep "n=" >> $out/ab_penbuf.log
  timeout 120 python scripts/probe_gls.py 100 1024 2.0 0 weight 2>&1 | grep "n=" >> $out/ab_penbuf.log
  timeout 120 python scripts/probe_gls.py 50 2048 1.0 0 model 2>&1 | grep "n=" >> $out/ab_penbuf.log
  timeout 120 python scripts/probe_gls.py 200 2048 2.0 0 model 2>&1 | grep "n=" >> $out/ab_penbuf.log
  timeout 120 python scripts/probe_gls.py 20 8192 1.0 0 model 2>&1 | grep "n=" >> $out/ab_penbuf.log
done
done
cat $out/ab_penbuf.log
