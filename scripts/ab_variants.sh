cd $GRAFT_REPO_ROOT
for v in _prev _kn _prev _kn; do
  export GNNGLS_HIP_SO=$PWD/gnngls_amd/libgnngls_hip$v.so
  echo "=== variant '$v'"
  timeout 120 python scripts/probe_gls.py 100 1024 2.0 0 noise 2>&1 | grep "n="
  timeout 120 python scripts/probe_gls.py 100 1024 2.0 0 weight 2>&1 | grep "n="
  timeout 120 python scripts/probe_gls.py 50 128 1.0 0 noise 2>&1 | grep "n="
done
