cd $GRAFT_REPO_ROOT
out=gpurun_out/r04d; mkdir -p $out
for v in stamps_nodual stamps_dual; do
  export GNNGLS_HIP_SO=$PWD/gnngls_amd/libgnngls_hip_$v.so
  for g in noise model weight; do
    echo "=== $v guide=$g" >> $out/stamps_dual.log
    timeout 120 python scripts/probe_gls_stamps.py 20 1000 $g 2>&1 | grep -v "^SIMD\|descent, per" >> $out/stamps_dual.log
  done
done
cat $out/stamps_dual.log
