cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -8
python scripts/forward_error_fixtures.py 100 200 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_forward_error_fixtures.txt
GNNGLS_FFN_FP32=1 python scripts/forward_error_fixtures.py 100 200 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_forward_error_fixtures_fp32pipe.txt
tail -3 gpurun_out/r06_forward_error_fixtures.txt
python __graft_entry__.py smoke 2>&1 | tail -2
