cd $GRAFT_REPO_ROOT
out=gpurun_out/r04l; mkdir -p $out
for rep in 1 2 3; do
for v in _oldgat ""; do
  export GNNGLS_HIP_SO=$PWD/gnngls_amd/libgnngls_hip$v.so
  echo "=== variant '$v'" >> $out/ab_gat_max.log
  timeout 120 python scripts/probe_forward.py 100 1024 3 2>&1 | grep -E "gat_rows|total" >> $out/ab_gat_max.log
  timeout 120 python scripts/probe_forward.py 200 256 3 2>&1 | grep -E "gat_rows|total" >> $out/ab_gat_max.log
  timeout 120 python scripts/probe_forward.py 50 2048 3 2>&1 | grep -E "gat_rows|total" >> $out/ab_gat_max.log
done
done
cat $out/ab_gat_max.log
