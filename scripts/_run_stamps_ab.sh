cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gls_gpu.py tests/test_gls_fuzz_gpu.py tests/test_search_progress_gpu.py -m gpu -q -x 2>&1 | tail -2
for v in stamps; do
for g in model noise weight; do
  echo "=== $v $g"
  GNNGLS_HIP_SO=$PWD/gnngls_amd/libgnngls_hip_$v.so timeout 200 python scripts/probe_gls_stamps.py 100 1024 $g 2>&1 | grep -v amdgpu.ids | grep "outer iters\|descent\|quiet\|cycles per pert"
done
done
for rep in 1 2; do
for v in _prev _new; do
  export GNNGLS_HIP_SO=$PWD/gnngls_amd/libgnngls_hip$v.so
  echo "=== variant '$v'"
  timeout 120 python scripts/probe_gls.py 100 1024 2.0 0 model 2>&1 | grep "n="
  timeout 120 python scripts/probe_gls.py 100 1024 2.0 0 weight 2>&1 | grep "n="
  timeout 120 python scripts/probe_gls.py 100 1024 2.0 0 noise 2>&1 | grep "n="
  timeout 120 python scripts/probe_gls.py 80 1024 2.0 0 model 2>&1 | grep "n="
  timeout 120 python scripts/probe_gls.py 120 512 2.0 0 model 2>&1 | grep "n="
done
done
