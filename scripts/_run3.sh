cd $GRAFT_REPO_ROOT
PMC_GUIDE=model bash scripts/pmc_gls.sh r05_pmc > gpurun_out/pmc_r05.log 2>&1
bash scripts/measure_round.sh r05a > gpurun_out/measure_r05a.log 2>&1
tail -3 gpurun_out/pmc_r05.log; ls gpurun_out/r05a; head -c 600 gpurun_out/r05a/bench.err
