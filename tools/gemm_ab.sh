cd $GRAFT_REPO_ROOT; export PYTHONPATH=$GRAFT_REPO_ROOT
for v in product bk64 bk16 mw3; do
  if [ "$v" != product ]; then export TVAE_LIB_OVERRIDE=$GRAFT_REPO_ROOT/transformer_vae_amd/variants/libtvae_$v.so; else unset TVAE_LIB_OVERRIDE; fi
  echo "== $v"
  OUT=$GRAFT_REPO_ROOT/gpurun_out/gemm_ab_$v; rm -rf $OUT; mkdir -p $OUT
  (cd /tmp && TMPDIR=/tmp timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/tools/bench_override.py > $OUT.log 2>&1)
  python3 - $OUT <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
tot=0
for r in csv.DictReader(open(f)):
    if "gemm" in r["Name"]:
        tot+=float(r["TotalDurationNs"]); print("   %-70s %3s %9.1f us" % (r["Name"][40:110], r["Calls"], float(r["AverageNs"]) / 1e3))
print("   gemm total per step %.2f ms" % (tot/4/1e6))
PY
done
