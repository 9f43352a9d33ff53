#!/bin/bash
# GPU session 3b of round 5: PMC passes of the headline launches (kernel-only), A/B of the fused kernel's issue priority
# (kernel alone and the CLI's -F / -F -D on a 4x list), the host-memory-load probe of the host decoder
set -u
O=gpurun_out/s3
mkdir -p $O
export TMPDIR=/tmp
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" \
           "GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i + 1))
    timeout -k 10 240 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pmc_$i -- python3 bench.py --kernel-only --steps 100 --warmup 20 --preheat 100 > $O/pmc_$i.log 2>&1 || { tail -5 $O/pmc_$i.log; exit 1; }
    echo "pmc $i done"
done
python3 tools/pmc_summary.py r05 $O > $O/pmc_summary.txt 2>&1
cp profiles/r05_pmc.json profiles/hbm_traffic.json $O/ 2>/dev/null
cat $O/pmc_summary.txt
python3 tools/ab_kernel.py phnrec_amd/lib/ab/libr5base.so phnrec_amd/lib/libphnrec_lcrc.so 8192 > $O/ab8192_prio.txt 2>&1 || exit 1
python3 tools/ab_kernel.py phnrec_amd/lib/ab/libr5base.so phnrec_amd/lib/libphnrec_lcrc.so 4096 > $O/ab4096_prio.txt 2>&1 || exit 1
cat $O/ab8192_prio.txt
python3 tools/ab_cli_list.py phnrec_amd/lib/ab/libr5base.so - 4 3 "-F" "-F -D" "-E -D" > $O/ab_cli_prio.txt 2>&1 || exit 1
cat $O/ab_cli_prio.txt
TIMELINE_REPS=4 python3 tools/cli_timeline.py 10000 -F > $O/timeline_x4.txt 2>&1
TIMELINE_REPS=4 python3 tools/cli_timeline.py 10000 -F -D >> $O/timeline_x4.txt 2>&1
TIMELINE_REPS=4 python3 tools/cli_timeline.py 10000 -E -D >> $O/timeline_x4.txt 2>&1
cat $O/timeline_x4.txt
# contexts per GPU with the decoder on the device: 3 (default) against 4 and 5
python3 tools/ab_cli_list.py - env:PHNREC_CTX_PER_GPU=4 4 2 "-F -D" "-E -D" > $O/ab_ctx4.txt 2>&1
python3 tools/ab_cli_list.py - env:PHNREC_CTX_PER_GPU=5 4 2 "-F -D" "-E -D" > $O/ab_ctx5.txt 2>&1
python3 tools/ab_cli_list.py - env:PHNREC_CTX_PER_GPU=4 4 2 "-F -D -b 32768" "-E -D -b 32768" > $O/ab_ctx4_b32k.txt 2>&1
cat $O/ab_ctx4.txt $O/ab_ctx5.txt $O/ab_ctx4_b32k.txt
./tools/ubench/host_mem_load 8 8 2 > $O/host_mem_load.txt 2>&1
./tools/ubench/host_mem_load 4 12 2 0 45.6 91.2 182.4 -1 >> $O/host_mem_load.txt 2>&1
cat $O/host_mem_load.txt
