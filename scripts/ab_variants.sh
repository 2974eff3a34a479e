cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_train_gpu.py tests/test_pipeline_gpu.py -m gpu -q -x 2>&1 | tail -3
for hs in 8 4; do
  for cfg in "100 1024" "50 2048" "200 256" "20 1000"; do
    echo "=== heads per workgroup $hs, TSP $cfg"
    GNNGLS_GAT_HEADS=$hs timeout 120 python scripts/probe_forward.py $cfg 3 2>&1 | grep -E "gat_rows|forward total"
  done
done
