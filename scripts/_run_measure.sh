cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r06a
timeout 300 python bench.py --steps 1 --warmup 1 --batch 64 --time_limit 0.5 --cpu_cores 8 --no_gap_bracket > gpurun_out/r06a/small.json 2> gpurun_out/r06a/small.err || { echo SMALL BENCH FAILED; tail -20 gpurun_out/r06a/small.err; exit 1; }
python scripts/iteration_rates.py gpurun_out/r06a/r06_iteration_rates.json > gpurun_out/r06a/iteration_rates.log 2>&1; tail -4 gpurun_out/r06a/iteration_rates.log
bash scripts/measure_round.sh r06a > gpurun_out/r06a/measure.log 2>&1; tail -3 gpurun_out/r06a/measure.log
PMC_GUIDE=model bash scripts/pmc_gls.sh r06_pmc > gpurun_out/r06a/pmc.log 2>&1
PMC_N=200 PMC_B=256 PMC_GUIDE=model bash scripts/pmc_gls.sh r06_pmc_tsp200 > gpurun_out/r06a/pmc200.log 2>&1
bash scripts/pmc_forward.sh r06_pmc_forward > gpurun_out/r06a/pmc_forward.log 2>&1
ls gpurun_out/r06a gpurun_out/r06_pmc gpurun_out/r06_pmc_tsp200 gpurun_out/r06_pmc_forward
