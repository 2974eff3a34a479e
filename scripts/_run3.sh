# last part of the round-5 measurement set: the other BASELINE sizes and the search kernel's PMC passes (GPU box, through gpurun)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r05d
python bench.py --n 50 --batch 128 --steps 1 --warmup 1 --no_cpu_baseline > gpurun_out/r05d/r05d_bench_tsp50x128.json 2>> gpurun_out/r05d/bench.err
python bench.py --n 200 --batch 256 --steps 1 --warmup 1 --no_cpu_baseline > gpurun_out/r05d/r05d_bench_tsp200x256.json 2>> gpurun_out/r05d/bench.err
python bench.py --n 50 --total_instances 2048 --steps 1 --warmup 1 --no_cpu_baseline > gpurun_out/r05d/r05d_bench_tsp50x2048.json 2>> gpurun_out/r05d/bench.err
python bench.py --n 20 --batch 1000 --steps 1 --warmup 1 --no_cpu_baseline --exact_gap > gpurun_out/r05d/r05d_bench_tsp20x1000_exact_gap.json 2>> gpurun_out/r05d/bench.err
PMC_GUIDE=model bash scripts/pmc_gls.sh r05_pmc > gpurun_out/r05d/pmc_gls.log 2>&1; tail -2 gpurun_out/r05d/pmc_gls.log
ls gpurun_out/r05d
