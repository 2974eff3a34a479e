# round 4, GPU call 3: dual half-wave guided scans (single-wavefront workgroups, n <= 33): exactness, then A/B
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/r04c; mkdir -p $out
timeout 1200 python -m pytest tests/test_gls_gpu.py tests/test_gls_fuzz_gpu.py tests/test_search_progress_gpu.py -m gpu -q -x 2>&1 | tail -3 > $out/pytest_dual.log
cat $out/pytest_dual.log
for rep in 1 2; do
for v in _nodual ""; do
  export GNNGLS_HIP_SO=$PWD/gnngls_amd/libgnngls_hip$v.so
  echo "=== variant '$v'" >> $out/ab_dual_o2a.log
  timeout 120 python scripts/probe_gls.py 20 1000 1.0 0 weight 2>&1 | grep "n=" >> $out/ab_dual_o2a.log
  timeout 120 python scripts/probe_gls.py 20 1000 1.0 0 model 2>&1 | grep "n=" >> $out/ab_dual_o2a.log
  timeout 120 python scripts/probe_gls.py 20 1000 1.0 0 noise 2>&1 | grep "n=" >> $out/ab_dual_o2a.log
  timeout 120 python scripts/probe_gls.py 30 1000 1.0 0 model 2>&1 | grep "n=" >> $out/ab_dual_o2a.log
  timeout 120 python scripts/probe_gls.py 12 1000 1.0 0 model 2>&1 | grep "n=" >> $out/ab_dual_o2a.log
  timeout 120 python scripts/probe_gls.py 20 4096 1.0 0 model 2>&1 | grep "n=" >> $out/ab_dual_o2a.log
done
done
cat $out/ab_dual_o2a.log
