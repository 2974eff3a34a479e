cd $GRAFT_REPO_ROOT
out=gpurun_out/r04p; mkdir -p $out
timeout 1200 python -m pytest tests/test_gls_gpu.py tests/test_gls_fuzz_gpu.py tests/test_search_progress_gpu.py -m gpu -q -x 2>&1 | tail -3 > $out/pytest_subst.log
cat $out/pytest_subst.log
for rep in 1 2; do
for v in _nosubst ""; do
  export GNNGLS_HIP_SO=$PWD/gnngls_amd/libgnngls_hip$v.so
  echo "=== variant '$v'" >> $out/ab_team_subst.log
  timeout 120 python scripts/probe_gls.py 200 256 2.0 0 model 2>&1 | grep "n=" >> $out/ab_team_subst.log
  timeout 120 python scripts/probe_gls.py 200 256 2.0 0 weight 2>&1 | grep "n=" >> $out/ab_team_subst.log
  timeout 120 python scripts/probe_gls.py 200 256 2.0 0 noise 2>&1 | grep "n=" >> $out/ab_team_subst.log
  timeout 120 python scripts/probe_gls.py 150 256 2.0 0 model 2>&1 | grep "n=" >> $out/ab_team_subst.log
  timeout 120 python scripts/probe_gls.py 100 1024 2.0 0 model 2>&1 | grep "n=" >> $out/ab_team_subst.log
done
done
cat $out/ab_team_subst.log
