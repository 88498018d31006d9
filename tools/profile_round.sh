#!/bin/bash
# Regenerates the judged profile summaries of a round on the GPU box:
#   bash tools/profile_round.sh r01        (writes gpurun_out/prof/<tag>_*; copy into profiles/)
# Kernel trace and every PMC group are separate rocprofv3 runs (never combined with sys traces).
TAG=${1:-r05}
R=$(pwd)
OUT=$R/gpurun_out/prof
mkdir -p $OUT
export TMPDIR=/tmp
finddb() { find "$1" -name '*.db' | head -1; }
SHORT="--steps 64 --batch 32 --min-seconds 0 --no-cpu-baseline --no-host-streamed --no-alt-modes --no-kalman-roofline --no-config3 --no-extra-configs --no-eval-png"
LIGHT="--no-cpu-baseline --no-host-streamed --no-alt-modes --no-extra-configs --no-eval-png"   # the c3 line without the blocks that run other configs
( cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT/kt -- python $R/bench.py $LIGHT --detail $OUT/${TAG}_bench_under_rocprof_detail.json > $OUT/${TAG}_bench_under_rocprof.json 2> $OUT/kt.err )
python tools/rocpd_stats.py "$(finddb $OUT/kt)" $OUT/${TAG}_kernel_stats.csv > /dev/null
# same trace with the two towers serialised on one stream: kernel durations without the
# overlap of the two-stream schedule, directly comparable with bench.py's isolated launches
( cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT/kt1 -- python $R/bench.py --one-stream $LIGHT --detail $OUT/${TAG}_bench_under_rocprof_one_stream_detail.json > $OUT/${TAG}_bench_under_rocprof_one_stream.json 2> $OUT/kt1.err )
python tools/rocpd_stats.py "$(finddb $OUT/kt1)" $OUT/${TAG}_kernel_stats_one_stream.csv > /dev/null
for C in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && rocprofv3 --kernel-trace --pmc $C -d $OUT/$C -- python $R/bench.py $SHORT > /dev/null 2> $OUT/$C.err )
done
python tools/pmc_traffic.py "$(finddb $OUT/FETCH_SIZE)" "$(finddb $OUT/WRITE_SIZE)" $OUT/${TAG}_pmc_traffic.json > /dev/null
# the Kalman-scan ROOFLINE launch (S=256 x T=64, the shape bench.py's roofline_kalman times) in its own PMC passes
for C in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && rocprofv3 --kernel-trace --pmc $C -d $OUT/k$C -- python $R/tools/kalman_roofline.py > $OUT/${TAG}_kalman_roofline_under_pmc_$C.json 2> $OUT/k$C.err )
done
python tools/pmc_traffic.py "$(finddb $OUT/kFETCH_SIZE)" "$(finddb $OUT/kWRITE_SIZE)" $OUT/${TAG}_pmc_traffic.json \
    --only kalman_scan_kernel --suffix '@S=256,T=64' --into $OUT/${TAG}_pmc_traffic.json > /dev/null
python tools/pmc_traffic.py "$(finddb $OUT/kFETCH_SIZE)" "$(finddb $OUT/kWRITE_SIZE)" $OUT/${TAG}_pmc_traffic.json \
    --only kalman_fuse_kernel --suffix '@P=78643200' --into $OUT/${TAG}_pmc_traffic.json > /dev/null
# ... and SURVEY 8(d)'s default shape S=256 x T=256 in two more passes (same kernel name: it needs its own)
for C in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && rocprofv3 --kernel-trace --pmc $C -d $OUT/k256$C -- python $R/tools/kalman_roofline.py --T 256 > $OUT/${TAG}_kalman_roofline_T256_under_pmc_$C.json 2> $OUT/k256$C.err )
done
python tools/pmc_traffic.py "$(finddb $OUT/k256FETCH_SIZE)" "$(finddb $OUT/k256WRITE_SIZE)" $OUT/${TAG}_pmc_traffic.json \
    --only kalman_scan_kernel --suffix '@S=256,T=256' --into $OUT/${TAG}_pmc_traffic.json > /dev/null
cp $OUT/${TAG}_pmc_traffic.json $R/profiles/${TAG}_pmc_traffic.json   # the bench line below quotes these numbers (newest rNN file)
i=0
# (fourth group, round 5: the texture-addresser side -- TA_BUSY_avr = % of the kernel's time the address units are busy,
#  vector-memory instructions issued, TA cycles stalled by the cache -- for the F(4x4,3x3) kernel's issue-rate analysis)
for G in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "TA_BUSY_avr SQ_INSTS_VMEM_RD TA_ADDR_STALLED_BY_TC_CYCLES_sum"; do
  i=$((i+1))
  ( cd /tmp && rocprofv3 --kernel-trace --pmc $G -d $OUT/sq$i -- python $R/bench.py $SHORT > /dev/null 2> $OUT/sq$i.err )
done
python tools/pmc_sq.py $OUT/${TAG}_pmc_sq_counters.json "$(finddb $OUT/sq1)" "$(finddb $OUT/sq2)" "$(finddb $OUT/sq3)" "$(finddb $OUT/sq4)"
# the un-profiled bench lines (they quote the fresh PMC traffic copied to profiles/ above): the 256-frame default and,
# after the config-5 passes below, the driver's own command
python bench.py --no-extra-configs --detail $OUT/${TAG}_bench_final_detail.json > $OUT/${TAG}_bench_final.json 2> $OUT/bench.err
rm -rf $OUT/kt $OUT/kt1 $OUT/FETCH_SIZE $OUT/WRITE_SIZE $OUT/kFETCH_SIZE $OUT/kWRITE_SIZE $OUT/k256FETCH_SIZE $OUT/k256WRITE_SIZE $OUT/sq1 $OUT/sq2 $OUT/sq3 $OUT/sq4
# ---- BASELINE config 5 (960x540, fp16 convs + fp16 activations, fp32 Kalman): kernel trace + the same PMC passes ----
C5="--config c5 --no-cpu-baseline --min-seconds 0"
( cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT/c5kt -- python $R/bench.py $C5 --detail $OUT/${TAG}_c5_bench_under_rocprof_detail.json > $OUT/${TAG}_c5_bench_under_rocprof.json 2> $OUT/c5kt.err )
python tools/rocpd_stats.py "$(finddb $OUT/c5kt)" $OUT/${TAG}_c5_kernel_stats.csv > /dev/null
# the same trace with the two towers serialised on one stream (VERDICT r4 Next #1a): per-kernel averages without the other
# stream's kernels sharing the CUs -- the file the c5 line's roofline.frac (isolated launches, HIP events) is recomputed from
( cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT/c5kt1 -- python $R/bench.py $C5 --one-stream --detail $OUT/${TAG}_c5_bench_under_rocprof_one_stream_detail.json > $OUT/${TAG}_c5_bench_under_rocprof_one_stream.json 2> $OUT/c5kt1.err )
python tools/rocpd_stats.py "$(finddb $OUT/c5kt1)" $OUT/${TAG}_c5_kernel_stats_one_stream.csv > /dev/null
# config 2 (single frame): kernel trace of the latency bench -- the per-layer batch-1 table of the line is recomputed from it
( cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT/c2kt -- python $R/bench.py --config c2 --min-seconds 0.5 --detail $OUT/${TAG}_c2_bench_under_rocprof_detail.json > $OUT/${TAG}_c2_bench_under_rocprof.json 2> $OUT/c2kt.err )
python tools/rocpd_stats.py "$(finddb $OUT/c2kt)" $OUT/${TAG}_c2_kernel_stats.csv > /dev/null
for C in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && rocprofv3 --kernel-trace --pmc $C -d $OUT/c5$C -- python $R/bench.py $C5 > /dev/null 2> $OUT/c5$C.err )
done
python tools/pmc_traffic.py "$(finddb $OUT/c5FETCH_SIZE)" "$(finddb $OUT/c5WRITE_SIZE)" $OUT/${TAG}_c5_pmc_traffic.json > /dev/null
i=0
for G in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  ( cd /tmp && rocprofv3 --kernel-trace --pmc $G -d $OUT/c5sq$i -- python $R/bench.py $C5 > /dev/null 2> $OUT/c5sq$i.err )
done
python tools/pmc_sq.py $OUT/${TAG}_c5_pmc_sq_counters.json "$(finddb $OUT/c5sq1)" "$(finddb $OUT/c5sq2)" "$(finddb $OUT/c5sq3)"
cp $OUT/${TAG}_c5_pmc_traffic.json $R/profiles/${TAG}_c5_pmc_traffic.json    # quoted by the c5 bench line below
python bench.py --config c5 --detail $OUT/${TAG}_bench_c5_detail.json > $OUT/${TAG}_bench_c5.json 2> $OUT/bench_c5.err
python bench.py --config c2 --detail $OUT/${TAG}_bench_c2_detail.json > $OUT/${TAG}_bench_c2.json 2> $OUT/bench_c2.err
python bench.py --gpus 1 --steps 20 --warmup 5 --detail $OUT/${TAG}_bench_driver_command_detail.json > $OUT/${TAG}_bench_driver_command.json 2> $OUT/bench_driver.err
rm -rf $OUT/c5kt $OUT/c5kt1 $OUT/c2kt $OUT/c5FETCH_SIZE $OUT/c5WRITE_SIZE $OUT/c5sq1 $OUT/c5sq2 $OUT/c5sq3
ls -la $OUT
