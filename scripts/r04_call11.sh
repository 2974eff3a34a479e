cd $GRAFT_REPO_ROOT
out=gpurun_out/r04k; mkdir -p $out
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_train_gpu.py tests/test_pipeline_gpu.py -m gpu -q -x 2>&1 | tail -3 > $out/pytest_model.log
cat $out/pytest_model.log
for rep in 1 2; do
timeout 120 python scripts/probe_forward.py 100 1024 3 2>&1 | grep -E "gat_rows|gemm_fc|ffn_fused|total" >> $out/forward_gat_max.log
timeout 120 python scripts/probe_forward.py 200 256 3 2>&1 | grep -E "gat_rows|total" >> $out/forward_gat_max.log
timeout 120 python scripts/probe_forward.py 50 2048 3 2>&1 | grep -E "gat_rows|total" >> $out/forward_gat_max.log
done
cat $out/forward_gat_max.log
