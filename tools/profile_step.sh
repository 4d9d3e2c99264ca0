# Round evidence for the 1024 step (run on the GPU box through gpurun):  bash tools/profile_step.sh [tag]
#   1. rocprofv3 --kernel-trace --stats of bench.py (5 graph steps)           -> <tag>_bench1024_kernel_stats.csv
#   2. rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, two SEPARATE passes with --kernel-trace only
#      (MI355X_MICROARCH.md, HBM section: the two counters do not fit one pass)  -> per-kernel-family traffic table
#   3. tools/traffic_table.py joins both with the algorithmic bytes of ops.Tracer (bench line's per_kernel_eager_step)
# Output under gpurun_out/<tag>/; copy the summaries you keep into profiles/.
TAG=${1:-r04}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp
rocprofv3 --kernel-trace --stats -d $O/stats -o out --output-format csv -- python3 $R/bench.py --steps 5 --warmup 1 --no-ae --no-trace --no-cpu-baseline > $O/bench_profiled.json 2> $O/bench_profiled.log || { tail -5 $O/bench_profiled.log; exit 1; }
echo "stats pass done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch -o out --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-ae --no-trace --no-cpu-baseline > $O/fetch.json 2> $O/fetch.log || { tail -5 $O/fetch.log; exit 1; }
echo "FETCH_SIZE pass done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write -o out --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-ae --no-trace --no-cpu-baseline > $O/write.json 2> $O/write.log || { tail -5 $O/write.log; exit 1; }
echo "WRITE_SIZE pass done"
cd $R && python3 tools/traffic_table.py $O $TAG
cp $O/stats/*kernel_stats.csv $O/${TAG}_bench1024_kernel_stats.csv 2>/dev/null || cp $O/stats/*/*kernel_stats.csv $O/${TAG}_bench1024_kernel_stats.csv
# keep only the summaries (the raw traces are hundreds of MB)
rm -rf $O/stats $O/fetch/*/*_kernel_trace.csv $O/write/*/*_kernel_trace.csv
ls -la $O
