cd $GRAFT_REPO_ROOT
export GNNGLS_HIP_SO=$PWD/gnngls_amd/libgnngls_hip_base.so
echo "=== base"
timeout 120 python scripts/probe_gls.py 20 1000 1.0 0 noise 2>&1 | grep "n="
timeout 120 python scripts/probe_gls.py 50 128,2048 1.0 0 noise 2>&1 | grep "n="
timeout 120 python scripts/probe_gls.py 200 256 2.0 0 noise 2>&1 | grep "n="
for v in _w8 _w4; do
  export GNNGLS_HIP_SO=$PWD/gnngls_amd/libgnngls_hip$v.so
  echo "=== variant '$v'"
  for thr in 64 128 256; do
    timeout 120 python scripts/probe_gls.py 20 1000 1.0 -2 noise $thr 2>&1 | grep "n="
  done
  for thr in 128 256 512; do
    timeout 120 python scripts/probe_gls.py 50 128,2048 1.0 -2 noise $thr 2>&1 | grep "n="
  done
done
export GNNGLS_HIP_SO=$PWD/gnngls_amd/libgnngls_hip_w8.so
timeout 120 python scripts/probe_gls.py 200 256 2.0 0 noise 2>&1 | grep "n="
