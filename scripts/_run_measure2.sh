cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
bash scripts/measure_round.sh r06b > gpurun_out/r06b_measure.log 2>&1; tail -3 gpurun_out/r06b_measure.log
