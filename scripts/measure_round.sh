# Measurement set of a round (run on the GPU box through gpurun): headline bench line, rocprofv3 kernel stats of the same
# command, other BASELINE sizes.  Usage: bash scripts/measure_round.sh r04a [quick]
tag=$1; quick=$2
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/$tag
python bench.py > gpurun_out/$tag/${tag}_bench.json 2> gpurun_out/$tag/bench.err
# under the profiler: no host-side worker pools started from the GPU-initialised parent (Held-Karp bracket, CPU baseline)
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag/prof -o ${tag} -- python3 bench.py --steps 1 --warmup 0 --no_cpu_baseline --no_gap_bracket --no_iso_quality > gpurun_out/$tag/${tag}_bench_under_rocprof.json 2> gpurun_out/$tag/rocprof.err
find gpurun_out/$tag/prof -name "*kernel_stats.csv" -exec cp {} gpurun_out/$tag/${tag}_kernel_stats.csv \;
rm -rf gpurun_out/$tag/prof
if [ -z "$quick" ]; then
python bench.py --n 50 --batch 128 --steps 1 --warmup 1 --no_cpu_baseline > gpurun_out/$tag/${tag}_bench_tsp50x128.json 2>> gpurun_out/$tag/bench.err
python bench.py --n 200 --batch 256 --steps 1 --warmup 1 --no_cpu_baseline > gpurun_out/$tag/${tag}_bench_tsp200x256.json 2>> gpurun_out/$tag/bench.err
python bench.py --n 50 --total_instances 2048 --steps 1 --warmup 1 --no_cpu_baseline > gpurun_out/$tag/${tag}_bench_tsp50x2048.json 2>> gpurun_out/$tag/bench.err
python bench.py --n 20 --batch 1000 --steps 1 --warmup 1 --no_cpu_baseline --exact_gap > gpurun_out/$tag/${tag}_bench_tsp20x1000_exact_gap.json 2>> gpurun_out/$tag/bench.err
fi
ls -la gpurun_out/$tag
