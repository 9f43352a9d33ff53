#!/bin/bash
# Regenerates everything under profiles/ that a round reports (run on the GPU box from the repo root):
#   make -C phnrec_amd/csrc stamps         (HERE, before the call: the stamped library travels with the snapshot; round 3's
#                                           phase_stamps.txt was a traceback of a stale one)
#   tools/refresh_profiles.sh r04          -> gpurun_out/refresh_r04/...   (tools/collect_profiles.sh copies the summaries)
# rocprofv3 is always given `python3 <script>` directly (no shell hop), PMC passes are separate runs
# with --kernel-trace only.  Every step writes a file under $OUT as it ends (progress for the harness).
set -u
TAG=${1:-r06}
PART=${2:-all}     # a | b | c | all: the run fits gpurun's 20-minute limit in three parts
OUT=gpurun_out/refresh_$TAG
[ "$PART" = "a" -o "$PART" = "all" ] && rm -rf $OUT    # (gpurun merges results into the local gpurun_out/: clear the local copy before calling, too)
mkdir -p $OUT
export TMPDIR=/tmp
if [ "$PART" = "a" -o "$PART" = "all" ]; then
# stdout = the bounded line the driver parses (<= 6 KB, numbers only); --detail-out = the full record of the same run
python3 bench.py --detail-out $OUT/bench.json > $OUT/bench_line.json 2> $OUT/bench.err
python3 bench.py --gpus 1 --steps 20 --warmup 5 --detail-out $OUT/bench_driver_form.json > $OUT/bench_driver_form_line.json 2>> $OUT/bench.err      # the driver's invocation
# the N-rank form on this box's one GPU (two ranks mapped onto it: functional, labelled oversubscribed)
PHNREC_DEVICE_MAP=0,0 python3 bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu --no-extras --detail-out $OUT/bench_2ranks_one_gpu.json > $OUT/bench_2ranks_one_gpu_line.json 2>> $OUT/bench.err
echo "bench done"
# per-kernel summary: rocprofv3's own (every launch of the process) and the steady-state one (the 200 timed launches)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --steps 200 --warmup 100 --no-cpu --no-extras --list-files 0 --detail-out $OUT/bench_profiled.json > $OUT/bench_profiled_line.json 2> $OUT/stats.log
python3 tools/steady_kernel_stats.py $OUT/stats 200 $OUT/steady_kernel_stats.csv $OUT/bench_profiled.json
# the fused kernel's own dispatch rows (grid, workgroup, LDS, registers)
for f in $OUT/stats/*/*kernel_trace.csv; do head -1 $f > $OUT/kernel_trace_head.csv; grep -m 3 lcrc_fused_kernel $f >> $OUT/kernel_trace_head.csv; done
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" \
           "GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i + 1))
    rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/pmc_$i -- python3 bench.py --kernel-only --steps 100 --warmup 20 --preheat 100 > $OUT/pmc_$i.log 2>&1
done
echo "pmc done"
python3 tools/stamp_profile.py > $OUT/phase_stamps.txt 2>&1
# (the shipped plan since round 4: pairs of 16-frame workgroups per CU)
LCRC_BM=16 python3 tools/stamp_profile.py PHN_CZ_SPDAT_LCRC_N1500 8192 >> $OUT/phase_stamps.txt 2>&1
LCRC_BM=16 python3 tools/stamp_profile.py PHN_CZ_SPDAT_LCRC_N1500 4096 >> $OUT/phase_stamps.txt 2>&1
python3 tools/stamp_profile.py PHN_EN_TIMIT_LCRC_N500 8192 >> $OUT/phase_stamps.txt 2>&1
LCRC_BM=16 python3 tools/stamp_profile.py PHN_EN_TIMIT_LCRC_N500 4096 >> $OUT/phase_stamps.txt 2>&1
fi
[ "$PART" = "a" ] && { ls -R $OUT | head -60; exit 0; }
if [ "$PART" = "b" -o "$PART" = "all" ]; then
# launch sizes incl. the cut points of the launch plan (whole rounds + cheaper tail), with the split path and without
python3 tools/system_sweep.py 2048 3072 4096 4100 5120 6144 8192 10240 12288 16384 32768 > $OUT/system_sweep.txt 2>&1
SWEEP_NO_SPLIT=1 python3 tools/system_sweep.py 4100 6144 12288 > $OUT/system_sweep_fused_only.txt 2>&1
python3 tools/small_launch_sweep.py > $OUT/small_launch_sweep.txt 2>&1
python3 tools/traps_bench.py > $OUT/traps_bench.txt 2>&1
echo "sweeps done"
for i in 1 2 3; do ./tools/ubench/hip_startup; done > $OUT/startup_probe.txt 2>&1
for m in 0 1 2 3 4; do ./tools/ubench/hip_upload $m; done >> $OUT/startup_probe.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/wave_stats -- python3 tools/frontend_bench.py 400 > $OUT/frontend_bench.txt 2>&1      # (400 calls: all but the first ~50 at the steady clock)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/traps_stats -- python3 tools/traps_bench.py > $OUT/traps_prof.txt 2>&1
# split-f16 arithmetic (opt-in): both arithmetics side by side, its own kernel stats
python3 tools/split_f16_bench.py > $OUT/split_f16_bench.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/split_stats -- python3 tools/split_f16_bench.py 8192 > $OUT/split_prof.txt 2>&1
fi
[ "$PART" = "b" ] && { ls -R $OUT | head -60; exit 0; }
# where a list run's wall clock goes on the device (the configs[3] list four times over), the modes by contexts per GPU,
# the host decoder's CPU time by thread count and under the posterior stores of eight GPUs, pinned-memory reads
TIMELINE_REPS=4 python3 tools/cli_timeline.py 10000 -F > $OUT/cli_timeline.txt 2>&1
TIMELINE_REPS=4 python3 tools/cli_timeline.py 10000 -F -D >> $OUT/cli_timeline.txt 2>&1
TIMELINE_REPS=4 python3 tools/cli_timeline.py 10000 -E -D >> $OUT/cli_timeline.txt 2>&1
TIMELINE_REPS=4 python3 tools/cli_timeline.py 10000 >> $OUT/cli_timeline.txt 2>&1
python3 tools/cli_sweep.py 10000 "3,0;2,0;4,0" > $OUT/cli_sweep.txt 2>&1
python3 tools/host_decoder_probe.py > $OUT/host_decoder_probe.txt 2>&1
./tools/ubench/host_mem_load 8 8 2 > $OUT/host_mem_load.txt 2>&1
./tools/ubench/host_mem_load 4 12 2 0 45.6 91.2 182.4 -1 >> $OUT/host_mem_load.txt 2>&1
./phnrec_amd/bin/phnrec --selftest-gpu-ln > $OUT/selftest_gpu_ln.txt 2>&1
# the first milliseconds of a list: the workers' steps and the library's slow calls
TRACE_CHARS=5000 python3 tools/pipeline_trace.py > $OUT/pipeline_trace.txt 2>&1
./tools/ubench/pinned_read 256 16 > $OUT/pinned_read.txt 2>&1
./tools/ubench/pinned_read 256 1 >> $OUT/pinned_read.txt 2>&1
ls -R $OUT | head -80
