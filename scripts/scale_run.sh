#!/bin/bash
# The driver's scaling run, spelled out: bench.py at N = 1, 2, 4, 8 GPUs of ONE node, one process per GPU over RCCL
# (backend "nccl"), each N after the other.  Run from the repo root on an 8-GPU MI355X node:
#     bash scripts/scale_run.sh [weak|strong|tsp200] [steps] [warmup]
#   weak   (default) BASELINE configs[2] per GPU: rank r searches block r (1024 TSP100 instances), 10 s budget per instance
#   strong           BASELINE configs[3]: the fixed 10,000-instance TSP100 test set cut into N contiguous shards (test.py:59);
#                    8 GPUs: 1250 per rank = 2 rounds of 625 -> residency_utilisation 0.61, i.e. ~500 instances/s by
#                    arithmetic against 819 for the weak-scaling headline (config.residency_utilisation in the line)
#   tsp200           BASELINE configs[4]: 256 TSP200 instances per GPU (one 16-wave workgroup per CU)
# One JSON line per N goes to scale_out/<mode>_n<N>.json.  There is no collective on the data path: the only exchange is
# one gather of [instances, 5] fp64 per step (config.collectives_per_step), so the efficiency a reader computes from the
# per-N `value`s measures launch skew and the gather, nothing else.
set -e
mode=${1:-weak}; steps=${2:-2}; warmup=${3:-1}
case $mode in
  weak)   extra="" ;;
  strong) extra="--total_instances 10000" ;;
  tsp200) extra="--tsp_n 200 --batch 256" ;;
  *) echo "usage: $0 [weak|strong|tsp200] [steps] [warmup]"; exit 2 ;;
esac
export HSA_ENABLE_IPC_MODE_LEGACY=0          # dmabuf IPC (RCCL across processes needs it on this driver)
mkdir -p scale_out
for N in 1 2 4 8; do
  if [ "$N" = 1 ]; then
    python bench.py --gpus 1 --steps $steps --warmup $warmup $extra > scale_out/${mode}_n$N.json
  else
    python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29600 + N)) \
        bench.py --gpus $N --steps $steps --warmup $warmup $extra > scale_out/${mode}_n$N.json
  fi
  python - "$N" scale_out/${mode}_n$N.json <<'PY'
import json, sys
j = json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][-1])
print(f"N={sys.argv[1]} value={j['value']:.2f} {j['unit']} gap={j['mean_gap_pct']} rounds={j['config']['rounds_per_rank']} "
      f"util={j['config']['residency_utilisation']} collectives/step={j['config']['collectives_per_step']} backend={j['config']['backend']}")
PY
done
