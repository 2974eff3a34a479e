tag=$1; N=$2; B=$3; K=$4; GUIDE=$5; shift 5
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
out=gpurun_out/$tag; mkdir -p $out
rocprofv3 -L 2>/dev/null | grep -o "SQC_[A-Z_]*" | sort -u | tr '\n' ' ' | head -c 3000; echo
for lib in "$@"; do
  export GNNGLS_HIP_SO=$PWD/gnngls_amd/libgnngls_hip$lib.so
  C="SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAIT_IFETCH SQ_INSTS_VALU SQ_WAVE_CYCLES"
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $out/tmp -o x -- python3 scripts/probe_gls_fixed.py $N $B $K $GUIDE > $out/log_${lib}_I.txt 2>&1
  f=$(find $out/tmp -name "*counter_collection.csv" | head -1)
  echo "== lib '$lib': $(grep 'n=' $out/log_${lib}_I.txt)"; tail -3 $out/log_${lib}_I.txt | cut -c1-300
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
