cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 3000 python -m pytest tests -m gpu -q 2>&1 | tail -6 > gpurun_out/r06c_gpu_suite.txt; tail -3 gpurun_out/r06c_gpu_suite.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash scripts/measure_round.sh r06c > gpurun_out/r06c_measure.log 2>&1; tail -2 gpurun_out/r06c_measure.log
