#!/bin/bash
# Regenerates everything under profiles/ that a round reports (run on the GPU box from the repo root):
#   tools/refresh_profiles.sh r02          -> gpurun_out/refresh_r02/...   (copy the summaries into profiles/)
# rocprofv3 is always given `python3 <script>` directly (no shell hop), PMC passes are separate runs
# with --kernel-trace only.
set -u
TAG=${1:-r02}
OUT=gpurun_out/refresh_$TAG
rm -rf $OUT    # (gpurun merges results into the local gpurun_out/: clear the local copy before calling, too)
mkdir -p $OUT
export TMPDIR=/tmp
python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_form.json 2>> $OUT/bench.err      # the driver's invocation
echo "bench done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --steps 200 --warmup 100 --no-cpu --no-extras > $OUT/stats.log 2>&1
# the fused kernel's own dispatch rows (grid, workgroup, LDS, registers)
for f in $OUT/stats/*/*kernel_trace.csv; do head -1 $f > $OUT/kernel_trace_head.csv; grep -m 3 lcrc_fused_kernel $f >> $OUT/kernel_trace_head.csv; done
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" \
           "GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i + 1))
    rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/pmc_$i -- python3 bench.py --steps 5 --warmup 2 --preheat 0 --no-cpu --no-extras > $OUT/pmc_$i.log 2>&1
done
echo "pmc done"
python3 tools/stamp_profile.py > $OUT/phase_stamps.txt 2>&1
LCRC_BM=16 python3 tools/stamp_profile.py PHN_CZ_SPDAT_LCRC_N1500 4096 >> $OUT/phase_stamps.txt 2>&1
python3 tools/stamp_profile.py PHN_EN_TIMIT_LCRC_N500 8192 >> $OUT/phase_stamps.txt 2>&1
LCRC_BM=16 python3 tools/stamp_profile.py PHN_EN_TIMIT_LCRC_N500 4096 >> $OUT/phase_stamps.txt 2>&1
python3 tools/system_sweep.py 2048 4096 8192 32768 > $OUT/system_sweep.txt 2>&1
python3 tools/small_launch_sweep.py > $OUT/small_launch_sweep.txt 2>&1
python3 tools/traps_bench.py > $OUT/traps_bench.txt 2>&1
echo "sweeps done"
for u in mfma_rate valu_overlap load_issue cross_wave mfma_shape; do ./tools/ubench/$u; done > $OUT/ubench.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/wave_stats -- python3 tools/frontend_bench.py > $OUT/frontend_bench.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/traps_stats -- python3 tools/traps_bench.py > $OUT/traps_prof.txt 2>&1
# split-f16 arithmetic (opt-in): both arithmetics side by side, its own kernel stats / MFMA counters / phase stamps
python3 tools/split_f16_bench.py > $OUT/split_f16_bench.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/split_stats -- python3 tools/split_f16_bench.py 8192 > $OUT/split_prof.txt 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $OUT/split_pmc -- python3 tools/split_f16_bench.py 8192 > $OUT/split_pmc.log 2>&1
rocprofv3 --pmc TA_BUSY_avr TA_BUSY_max GRBM_GUI_ACTIVE TCP_TOTAL_CACHE_ACCESSES_sum --kernel-trace --output-format csv -d $OUT/split_ta_pmc -- python3 tools/split_f16_bench.py 8192 > $OUT/split_ta_pmc.log 2>&1
LCRC_ARITH=1 python3 tools/stamp_profile.py > $OUT/split_phase_stamps.txt 2>&1
LCRC_ARITH=1 LCRC_BM=16 python3 tools/stamp_profile.py PHN_EN_TIMIT_LCRC_N500 4096 >> $OUT/split_phase_stamps.txt 2>&1
./tools/ubench/split_f16 > $OUT/split_ubench.txt 2>&1
python3 tools/cli_throughput.py 2000 > $OUT/cli_throughput.txt 2>&1
python3 tools/cli_throughput.py 10000 >> $OUT/cli_throughput.txt 2>&1
ls -R $OUT | head -60
