# PMC passes on the forward kernels (run on the GPU box through gpurun): counters in their own runs, kernel-trace only.
#   bash scripts/pmc_forward.sh r03_pmc_forward   -> gpurun_out/r03_pmc_forward/{mfma,valu,fetch,write}_counter_collection.csv
# Workload: predict_regret on 512 TSP100 instances (scripts/probe_forward.py 100 512 2), the shape of one forward chunk.
tag=$1
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/$tag
mkdir -p $out
run() {   # name, counters...
  name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $out/tmp_$name -o $name -- python3 scripts/probe_forward.py 100 512 2 > $out/$name.log 2>&1
  f=$(find $out/tmp_$name -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then grep -E "Counter_Name|ffn_fused|gat_rows_kernel|gemm_f32_kernel" "$f" > $out/${name}_counter_collection.csv; fi
  rm -rf $out/tmp_$name
  tail -2 $out/$name.log
}
rocprofv3 --list-avail 2>/dev/null | grep -o "SQ_[A-Z_0-9]*MFMA[A-Z_0-9]*" | sort -u > $out/mfma_counters_available.txt
run mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY
run valu SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE
run fetch FETCH_SIZE
run write WRITE_SIZE
ls -la $out
