cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/r04h; mkdir -p $out
PMC_GUIDE=model bash scripts/pmc_gls.sh r04_pmc > $out/pmc.log 2>&1
tail -5 $out/pmc.log
PMC_GUIDE=model PMC_N=200 PMC_B=256 bash scripts/pmc_gls.sh r04_pmc_tsp200 > $out/pmc200.log 2>&1
tail -3 $out/pmc200.log
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > $out/pytest.log
cat $out/pytest.log
