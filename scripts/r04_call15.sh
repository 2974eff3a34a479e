cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/r04o; mkdir -p $out
bash scripts/pmc_forward.sh r04_pmc_forward > $out/pmc_forward.log 2>&1
tail -3 $out/pmc_forward.log
bash scripts/measure_round.sh r04c > $out/measure.log 2>&1
tail -3 $out/measure.log
