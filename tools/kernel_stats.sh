# Per-kernel time of the 1024 step only (no PMC passes):  bash tools/kernel_stats.sh [tag]  -> gpurun_out/<tag>/ks.csv
TAG=${1:-mid}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/stats -o out --output-format csv -- python3 $R/bench.py --steps 5 --warmup 1 --no-ae --no-trace --no-cpu-baseline > $O/b.json 2> $O/b.log || { tail -5 $O/b.log; exit 1; }
cp $O/stats/*kernel_stats.csv $O/ks.csv 2>/dev/null || cp $O/stats/*/*kernel_stats.csv $O/ks.csv
rm -rf $O/stats
cut -c1-200 $O/b.json
