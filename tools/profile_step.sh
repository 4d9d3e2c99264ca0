cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final
mkdir -p $O
cd $R && python bench.py --steps 10 --warmup 2 > $O/bench_full.json 2> $O/bench_full.log || { tail -5 $O/bench_full.log; exit 1; }
cd /tmp
rocprofv3 --kernel-trace --stats -d $O/stats -o out --output-format csv -- python3 $R/bench.py --steps 5 --warmup 1 --no-ae --no-trace --no-cpu-baseline > $O/bench_profiled.json 2> $O/bench_profiled.log || { tail -5 $O/bench_profiled.log; exit 1; }
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch -o out --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-ae --no-trace --no-cpu-baseline > $O/fetch.json 2> $O/fetch.log || { tail -5 $O/fetch.log; exit 1; }
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write -o out --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-ae --no-trace --no-cpu-baseline > $O/write.json 2> $O/write.log || { tail -5 $O/write.log; exit 1; }
python3 - <<'PY'
import csv, os
O=os.environ.get("GRAFT_REPO_ROOT")+"/gpurun_out/final"
for n,c in (("fetch","FETCH_SIZE"),("write","WRITE_SIZE")):
    tot=0.0
    for r in csv.DictReader(open(f"{O}/{n}/out_counter_collection.csv")):
        if r["Counter_Name"]==c: tot+=float(r["Counter_Value"])
    print(n, c, "sum", tot)
PY
ls $O/stats
