cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/r04i; mkdir -p $out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > $out/pytest.log
cat $out/pytest.log
bash scripts/measure_round.sh r04b > $out/measure.log 2>&1
tail -3 $out/measure.log
