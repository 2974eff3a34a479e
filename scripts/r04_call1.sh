# round 4, GPU call 1: full GPU suite, A/B of the executed-evaluation counter, thread policy probe, first bench line
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/r04a; mkdir -p $out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $out/pytest.log
cat $out/pytest.log | tail -5
for rep in 1 2; do
for v in _noexec ""; do
  export GNNGLS_HIP_SO=$PWD/gnngls_amd/libgnngls_hip$v.so
  echo "=== variant '$v'" >> $out/ab_exec_counter.log
  timeout 120 python scripts/probe_gls.py 100 1024 2.0 0 model 2>&1 | grep "n=" >> $out/ab_exec_counter.log
  timeout 120 python scripts/probe_gls.py 100 1024 2.0 0 noise 2>&1 | grep "n=" >> $out/ab_exec_counter.log
  timeout 120 python scripts/probe_gls.py 200 256 2.0 0 model 2>&1 | grep "n=" >> $out/ab_exec_counter.log
  timeout 120 python scripts/probe_gls.py 20 1000 1.0 0 weight 2>&1 | grep "n=" >> $out/ab_exec_counter.log
done
done
unset GNNGLS_HIP_SO
cat $out/ab_exec_counter.log
# workgroup size for n = 30 beyond the 128-VGPR residency (64-VGPR build): one wavefront (r03 policy) vs two (now)
for thr in 64 128; do
  timeout 120 python scripts/probe_gls.py 30 8192 1.0 -2 weight $thr 2>&1 | grep -E "n=|capacity" >> $out/ab_threads_tsp30x8192.log
done
cat $out/ab_threads_tsp30x8192.log
timeout 600 python bench.py --steps 2 --warmup 1 > $out/r04a_bench.json 2> $out/bench.err
tail -c 3000 $out/r04a_bench.json
