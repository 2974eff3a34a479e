# Round-end validation on the GPU box (through gpurun): the -m gpu suite in groups (each under its own timeout, so that one hang
# cannot eat the call), smoke(), per-second iteration rates for tests/test_perf_floor_gpu.py, the measurement set of the round.
# Usage: bash scripts/final_check.sh r05b
tag=${1:-r05b}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/${tag}_check; mkdir -p $out
grp() { name=$1; shift; timeout 900 python -m pytest "$@" -m gpu -x -q 2>&1 | tail -4 > $out/pytest_$name.log; echo "$name: $(tail -1 $out/pytest_$name.log)"; }
grp search tests/test_gls_gpu.py tests/test_gls_fuzz_gpu.py tests/test_search_progress_gpu.py tests/test_mirror_gpu.py
grp forward tests/test_model_gpu.py tests/test_torch_ops_gpu.py tests/test_n3_ingestion_gpu.py tests/test_pipeline_gpu.py
grp train tests/test_train_gpu.py
python scripts/iteration_rates.py $out/iteration_rates.json > $out/iteration_rates.log 2>&1; tail -2 $out/iteration_rates.log
grp bench tests/test_bench_gpu.py tests/test_perf_floor_gpu.py
python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; tail -1 $out/smoke.log
bash scripts/measure_round.sh $tag > $out/measure.log 2>&1
tail -2 $out/measure.log
