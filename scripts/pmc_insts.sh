# dynamic instruction counts of gls_kernel at a fixed iteration count, for two builds:  bash scripts/pmc_insts.sh tag n B iters guide lib1 lib2 ...
tag=$1; N=$2; B=$3; K=$4; GUIDE=$5; shift 5
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
out=gpurun_out/$tag; mkdir -p $out
for lib in "$@"; do
  export GNNGLS_HIP_SO=$PWD/gnngls_amd/libgnngls_hip$lib.so
  for pass in A B; do
    if [ $pass = A ]; then C="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_WAIT_INST_ANY"; else C="SQ_INSTS_BRANCH SQ_INSTS_VALU_TRANS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAVES GRBM_GUI_ACTIVE SQ_INST_CYCLES_SALU"; fi
    rocprofv3 --kernel-trace --pmc $C --output-format csv -d $out/tmp -o x -- python3 scripts/probe_gls_fixed.py $N $B $K $GUIDE > $out/log_${lib}_$pass.txt 2>&1
    f=$(find $out/tmp -name "*counter_collection.csv" | head -1)
    echo "== lib '$lib' pass $pass: $(grep 'n=' $out/log_${lib}_$pass.txt)"
    if [ -n "$f" ]; then python3 - "$f" <<'PY'
import csv, sys
acc = {}
for row in csv.DictReader(open(sys.argv[1])):
    if "gls_kernel" in row["Kernel_Name"]:
        acc[row["Counter_Name"]] = acc.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
print("   " + "  ".join(f"{k}={v:.4e}" for k, v in sorted(acc.items())))
PY
    fi
    rm -rf $out/tmp
  done
done
