# Per-kernel time of one config's step only (no PMC passes):  bash tools/kernel_stats.sh [tag] [res]  -> gpurun_out/<tag>/ks[_res].csv
TAG=${1:-mid}
RES=${2:-1024}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/stats -o out --output-format csv -- python3 $R/bench.py --res $RES --steps 5 --warmup 1 --no-ae --no-trace --no-cpu-baseline > $O/b.json 2> $O/b.log || { tail -5 $O/b.log; exit 1; }
KS=$O/ks.csv; [ "$RES" != 1024 ] && KS=$O/${TAG}_bench${RES}_kernel_stats.csv
cp $O/stats/*kernel_stats.csv $KS 2>/dev/null || cp $O/stats/*/*kernel_stats.csv $KS
rm -rf $O/stats
cut -c1-200 $O/b.json
