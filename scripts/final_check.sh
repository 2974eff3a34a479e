cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/r04r; mkdir -p $out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > $out/pytest.log
cat $out/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; tail -2 $out/smoke.log
bash scripts/measure_round.sh r04d > $out/measure.log 2>&1
tail -2 $out/measure.log
timeout 900 python bench.py --gpus 1 --steps 5 --warmup 2 --no_gap_bracket > $out/bench_steps5.json 2> $out/bench_steps5.err
tail -c 600 $out/bench_steps5.json
