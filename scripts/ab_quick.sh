# same-box A/B of HIP library variants:  bash scripts/ab_quick.sh "_prev _e1 ''" [tests]
cd $GRAFT_REPO_ROOT
if [ -n "$2" ]; then timeout 1200 python -m pytest tests/test_gls_gpu.py tests/test_gls_fuzz_gpu.py tests/test_search_progress_gpu.py -m gpu -q -x 2>&1 | tail -2; fi
if [ -f gnngls_amd/libgnngls_hip_stamps.so ]; then
export GNNGLS_HIP_SO=$PWD/gnngls_amd/libgnngls_hip_stamps.so
timeout 120 python scripts/probe_gls_stamps.py 100 1024 model 2>&1 | grep -v amdgpu.ids | head -11
timeout 120 python scripts/probe_gls_stamps.py 20 1000 model 2>&1 | grep -v amdgpu.ids | head -9
fi
for rep in 1 2; do
for v in $1; do
  export GNNGLS_HIP_SO=$PWD/gnngls_amd/libgnngls_hip$v.so
  echo "=== variant '$v'"
  timeout 120 python scripts/probe_gls.py 100 1024 2.0 0 model 2>&1 | grep "n="
  timeout 120 python scripts/probe_gls.py 100 1024 2.0 0 weight 2>&1 | grep "n="
  timeout 120 python scripts/probe_gls.py 100 1024 2.0 0 noise 2>&1 | grep "n="
  timeout 120 python scripts/probe_gls.py 50 128 1.0 0 model 2>&1 | grep "n="
  timeout 120 python scripts/probe_gls.py 20 1000 1.0 0 model 2>&1 | grep "n="
  timeout 120 python scripts/probe_gls.py 150 256 2.0 0 model 2>&1 | grep "n="
  timeout 120 python scripts/probe_gls.py 200 256 2.0 0 model 2>&1 | grep "n="
done
done
