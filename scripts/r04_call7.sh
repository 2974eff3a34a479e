cd $GRAFT_REPO_ROOT
out=gpurun_out/r04g; mkdir -p $out
for rep in 1 2 3; do
for v in _lpr8 ""; do
  export GNNGLS_HIP_SO=$PWD/gnngls_amd/libgnngls_hip$v.so
  echo "=== variant '$v'" >> $out/ab_lpr7_b.log
  timeout 120 python scripts/probe_gls.py 100 1024 2.0 0 model 2>&1 | grep "n=" >> $out/ab_lpr7_b.log
  timeout 120 python scripts/probe_gls.py 50 128 1.0 0 model 2>&1 | grep "n=" >> $out/ab_lpr7_b.log
  timeout 120 python scripts/probe_gls.py 50 2048 1.0 0 model 2>&1 | grep "n=" >> $out/ab_lpr7_b.log
  timeout 120 python scripts/probe_gls.py 64 1024 1.0 0 model 2>&1 | grep "n=" >> $out/ab_lpr7_b.log
done
done
cat $out/ab_lpr7_b.log
