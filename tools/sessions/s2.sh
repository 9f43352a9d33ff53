#!/bin/bash
# GPU session 2 of round 5: GPU tests, the driver's bench form, PMC passes of the headline launches (kernel-only)
set -u
O=gpurun_out/s2
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 500 python -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1
rc=$?
echo "tests rc=$rc"; tail -3 $O/gputests.log
[ $rc -eq 0 ] || exit $rc
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench.err || exit 1
echo "bench done"
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" \
           "GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i + 1))
    timeout -k 10 240 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pmc_$i -- python3 bench.py --kernel-only --steps 100 --warmup 20 --preheat 100 > $O/pmc_$i.log 2>&1 || exit 1
    echo "pmc $i done"
done
python3 tools/pmc_summary.py r05 $O > $O/pmc_summary.txt 2>&1
cp profiles/r05_pmc.json profiles/hbm_traffic.json $O/ 2>/dev/null
cat $O/pmc_summary.txt
