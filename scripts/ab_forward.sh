# same-box A/B of HIP library variants on the forward:  bash scripts/ab_forward.sh "_prev _new" [tests]   (per-kernel-class device times of scripts/probe_forward.py 100 1024 3, twice per variant)
cd $GRAFT_REPO_ROOT
if [ -n "$2" ]; then timeout 1500 python -m pytest tests/test_model_gpu.py -m gpu -q -x 2>&1 | tail -3; fi
for rep in 1 2; do
for v in $1; do
  export GNNGLS_HIP_SO=$PWD/gnngls_amd/libgnngls_hip$v.so
  echo "=== variant '$v'"
  timeout 300 python scripts/probe_forward.py 100 1024 3 2>&1 | grep -E "ffn_fused|gat_rows|total|forward"
done
done
