# PMC passes on the search kernel (run on the GPU box through gpurun): counters in their own runs, kernel-trace only.
#   bash scripts/pmc_gls.sh r02_pmc        -> gpurun_out/r02_pmc/{fetch,write,lds,issue}_counter_collection.csv
tag=$1
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/$tag
mkdir -p $out
run() {   # name, counters...
  name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $out/tmp_$name -o $name -- python3 scripts/probe_gls.py 100 1024 2.0 0 noise > $out/$name.log 2>&1
  f=$(find $out/tmp_$name -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then grep -E "Counter_Name|gls_kernel" "$f" > $out/${name}_counter_collection.csv; fi
  rm -rf $out/tmp_$name
  grep "n=" $out/$name.log
}
run fetch FETCH_SIZE
run write WRITE_SIZE
run lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_ANY
run issue SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM
ls -la $out
