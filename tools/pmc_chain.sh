cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_chain; rm -rf $OUT; mkdir -p $OUT
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/$C -o t -- python3 tools/time_bwd_parts.py --reps 2 > $OUT/$C.json 2> $OUT/$C.err
done
find $OUT -name "*.csv" | head
python3 - <<'PY'
import csv,glob,collections
for C in ("FETCH_SIZE","WRITE_SIZE"):
    f=glob.glob("gpurun_out/pmc_chain/%s/**/*counter_collection.csv"%C, recursive=True)
    if not f: print("no csv for",C); continue
    agg=collections.defaultdict(lambda:[0,0.0])
    for r in csv.DictReader(open(f[0])):
        k=r["Kernel_Name"][:50]; agg[k][0]+=1; agg[k][1]+=float(r["Counter_Value"])
    for k,(n,v) in sorted(agg.items(), key=lambda kv:-kv[1][1])[:6]:
        print(C, "%-52s n=%3d per launch %.1f (counter units)"%(k,n,v/n))
PY
