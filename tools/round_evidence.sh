# Everything the round's numbers come from, on one box (run through gpurun):  bash tools/round_evidence.sh <tag>
#   <tag>_bench1024.json          default bench line (20 steps; roofline table, measured cpu_baseline + parity check)
#   <tag>_bench1024_guard.json    the same workload with DC_ARENA_GUARD=1 (scratch sentinels verified at full size)
#   <tag>_bench512.json / 256     the other released configs (parity-test cases; builder-run numbers)
#   tools/profile_step.sh         rocprofv3 kernel stats + PMC traffic table
#   tools/pmc_one_gemm.sh         PMC counters of single kernels: flash attention 32x5x9216, the fused level-0 kernels (one_fused.py)
# optional second argument: which part (bench | prof | small | pmc | all): gpurun limits one call to 20 minutes
TAG=${1:-r04}
PART=${2:-all}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
if [ "$PART" = all ] || [ "$PART" = bench ]; then
python bench.py --steps 20 --warmup 2 > $O/${TAG}_bench1024.json 2> $O/bench1024.log || { tail -5 $O/bench1024.log; exit 1; }
cp $O/${TAG}_bench1024.json $O/bench_full.json
grep -E "timed|AE|cpu_baseline" $O/bench1024.log
DC_ARENA_GUARD=1 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-trace > $O/${TAG}_bench1024_guard.json 2> $O/guard.log || { tail -5 $O/guard.log; exit 1; }
grep -E "timed|guard" $O/guard.log
python bench.py --res 512 --steps 20 --warmup 2 > $O/${TAG}_bench512.json 2> $O/bench512.log || { tail -5 $O/bench512.log; exit 1; }
grep -E "timed|cpu_baseline: oracle" $O/bench512.log
python bench.py --res 256 --steps 20 --warmup 2 > $O/${TAG}_bench256.json 2> $O/bench256.log || { tail -5 $O/bench256.log; exit 1; }
grep -E "timed|cpu_baseline: oracle" $O/bench256.log
fi
if [ "$PART" = all ] || [ "$PART" = prof ]; then
bash tools/profile_step.sh $TAG > $O/profile.log 2>&1 || { tail -5 $O/profile.log; exit 1; }
head -30 $O/${TAG}_traffic_by_kernel.md
fi
if [ "$PART" = all ] || [ "$PART" = small ]; then
# rocprofv3 kernel stats of the 512 / 256 configs' steps (the other released configs: parity-test cases, builder-run numbers)
bash tools/kernel_stats.sh $TAG 512 > $O/ks512.log 2>&1 || tail -5 $O/ks512.log
bash tools/kernel_stats.sh $TAG 256 > $O/ks256.log 2>&1 || tail -5 $O/ks256.log
AE_FRAMES=4 AE_DETAIL=1 AE_DETAIL_N=40 python tools/ae_profile.py > $O/${TAG}_ae_profile_4frames.txt 2>&1; DC_GEMM_PLAN=403 AE_FRAMES=4 python tools/ae_profile.py 2>&1 | grep "==" > $O/ae_tile.txt; echo "with the tile kernels only (DC_GEMM_PLAN=403: window / narrow conv kernels off):" >> $O/${TAG}_ae_profile_4frames.txt; cat $O/ae_tile.txt >> $O/${TAG}_ae_profile_4frames.txt
ls $O | grep -i "kernel_stats\|ae_profile"
fi
if [ "$PART" = all ] || [ "$PART" = pmc ]; then
# PMC counters (MFMA busy, wait / issue split, L1 pending stalls, L2 <-> memory requests) of single kernels
PMC_SCRIPT=one_flash.py bash tools/pmc_one_gemm.sh flash 32 5 9216 9216 > $O/pmc_flash.log 2>&1; python tools/pmc_summary.py $R/gpurun_out/pmc_flash > $O/${TAG}_pmc_flash.txt 2>&1; tail -12 $O/${TAG}_pmc_flash.txt
# the level-1 skip-connection conv [73728 x 640 x 17280] on the default plan (gemm_pipe320x16_kernel) and on the 8-wave kernels
bash tools/pmc_one_gemm.sh conv16 conv 1920 640 36 64 > $O/pmc_conv16.log 2>&1; python tools/pmc_summary.py $R/gpurun_out/pmc_conv16 > $O/${TAG}_pmc_conv16.txt 2>&1; tail -12 $O/${TAG}_pmc_conv16.txt
DC_GEMM_PLAN=0 bash tools/pmc_one_gemm.sh conv8w conv 1920 640 36 64 > $O/pmc_conv8w.log 2>&1; python tools/pmc_summary.py $R/gpurun_out/pmc_conv8w > $O/${TAG}_pmc_conv8w.txt 2>&1; tail -12 $O/${TAG}_pmc_conv8w.txt
# the GEGLU projections (level 1: ping-pong kernel, level 2: persistent 8-wave kernel) and the level-2 linear (gemm_persist<128>)
bash tools/pmc_one_gemm.sh geglu_l1 geglu 640 5120 36 64 > $O/pmc_geglu_l1.log 2>&1; python tools/pmc_summary.py $R/gpurun_out/pmc_geglu_l1 > $O/${TAG}_pmc_geglu_l1.txt 2>&1; tail -4 $O/${TAG}_pmc_geglu_l1.txt
bash tools/pmc_one_gemm.sh geglu_l2 geglu 1280 10240 18 32 > $O/pmc_geglu_l2.log 2>&1; python tools/pmc_summary.py $R/gpurun_out/pmc_geglu_l2 > $O/${TAG}_pmc_geglu_l2.txt 2>&1; tail -4 $O/${TAG}_pmc_geglu_l2.txt
bash tools/pmc_one_gemm.sh lin1280 lin 1280 1280 18 32 > $O/pmc_lin1280.log 2>&1; python tools/pmc_summary.py $R/gpurun_out/pmc_lin1280 > $O/${TAG}_pmc_persist128.txt 2>&1; tail -4 $O/${TAG}_pmc_persist128.txt
PMC_SCRIPT=one_xattn.py bash tools/pmc_one_gemm.sh xattn > $O/pmc_xattn.log 2>&1; python tools/pmc_summary.py $R/gpurun_out/pmc_xattn > $O/${TAG}_pmc_xattn.txt 2>&1; tail -4 $O/${TAG}_pmc_xattn.txt
# the AE's full-resolution N = 128 conv [2359296 x 128 x 1152]: the window kernel and (plan bit 8) the 256 x 128 tile kernel it replaced
bash tools/pmc_one_gemm.sh window128 conv 128 128 288 256 > $O/pmc_window128.log 2>&1; python tools/pmc_summary.py $R/gpurun_out/pmc_window128 > $O/${TAG}_pmc_window128.txt 2>&1; tail -4 $O/${TAG}_pmc_window128.txt
DC_GEMM_PLAN=275 bash tools/pmc_one_gemm.sh ae128tile conv 128 128 288 256 > $O/pmc_ae128tile.log 2>&1; python tools/pmc_summary.py $R/gpurun_out/pmc_ae128tile > $O/${TAG}_pmc_ae128_tile.txt 2>&1; tail -4 $O/${TAG}_pmc_ae128_tile.txt
for k in ff tconv lnlin linres; do
  PMC_SCRIPT=one_fused.py bash tools/pmc_one_gemm.sh $k $k > $O/pmc_$k.log 2>&1; python tools/pmc_summary.py $R/gpurun_out/pmc_$k > $O/${TAG}_pmc_$k.txt 2>&1; tail -8 $O/${TAG}_pmc_$k.txt
done
fi
