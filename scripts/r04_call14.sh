cd $GRAFT_REPO_ROOT
out=gpurun_out/r04n; mkdir -p $out
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_train_gpu.py -m gpu -q -x 2>&1 | tail -3 > $out/pytest_model.log
cat $out/pytest_model.log
for rep in 1 2; do
for v in _oldgat ""; do
  export GNNGLS_HIP_SO=$PWD/gnngls_amd/libgnngls_hip$v.so
  echo "=== variant '$v'" >> $out/ab_gat_elfuse.log
  timeout 120 python scripts/probe_forward.py 100 1024 3 2>&1 | grep -E "gat_rows|total" >> $out/ab_gat_elfuse.log
  timeout 120 python scripts/probe_forward.py 200 256 3 2>&1 | grep -E "gat_rows|total" >> $out/ab_gat_elfuse.log
  timeout 120 python scripts/probe_forward.py 50 2048 3 2>&1 | grep -E "gat_rows|total" >> $out/ab_gat_elfuse.log
  timeout 120 python scripts/probe_forward.py 20 4096 3 2>&1 | grep -E "gat_rows|total" >> $out/ab_gat_elfuse.log
done
done
cat $out/ab_gat_elfuse.log
