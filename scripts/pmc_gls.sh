# PMC passes on the search kernel (run on the GPU box through gpurun): counters in their own runs, kernel-trace only.
#   bash scripts/pmc_gls.sh r02_pmc        -> gpurun_out/r02_pmc/{fetch,write,lds,issue}_counter_collection.csv
#   PMC_N / PMC_B / PMC_GUIDE / PMC_TEAM / PMC_PASSES select another workload (default: TSP100 x 1024, noise guide, all four passes; bench.py's own guide: PMC_GUIDE=model);
#   the workload is recorded in workload.json, which scripts/pmc_summary.py copies into the summary
tag=$1
N=${PMC_N:-100}; B=${PMC_B:-1024}; GUIDE=${PMC_GUIDE:-noise}; TEAM=${PMC_TEAM:--1}; PASSES=${PMC_PASSES:-fetch write lds issue}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/$tag
mkdir -p $out
echo "{\"n\": $N, \"instances\": $B, \"guide\": \"$GUIDE\", \"seconds\": 2.0, \"team\": $TEAM}" > $out/workload.json
run() {   # name, counters...
  name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $out/tmp_$name -o $name -- python3 scripts/probe_gls.py $N $B 2.0 0 $GUIDE 0 $TEAM > $out/$name.log 2>&1
  f=$(find $out/tmp_$name -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then grep -E "Counter_Name|gls_kernel" "$f" > $out/${name}_counter_collection.csv; fi
  rm -rf $out/tmp_$name
  grep "n=" $out/$name.log
}
for pass in $PASSES; do
case $pass in
fetch) run fetch FETCH_SIZE ;;
write) run write WRITE_SIZE ;;
lds) run lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_ANY ;;
issue) run issue SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM ;;
sched) run sched SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_WAVE_CYCLES GRBM_GUI_ACTIVE ;;
esac
done
ls -la $out
