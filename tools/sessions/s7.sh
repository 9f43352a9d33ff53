#!/bin/bash
# GPU session 7 of round 5: launch order without the overlapped decoder at 2 / 3 / 4 contexts per GPU
set -u
O=gpurun_out/s7
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
python3 tools/ab_cli_list.py env:PHNREC_NO_OVERLAP=1,PHNREC_CTX_PER_GPU=2 env:PHNREC_CTX_PER_GPU=2 4 3 "-F -D" "-E -D" > $O/ab_ctx2_order_vs_both.txt 2>&1 || exit 1
cat $O/ab_ctx2_order_vs_both.txt
python3 tools/ab_cli_list.py env:PHNREC_NO_OVERLAP=1 env:PHNREC_NO_OVERLAP=1,PHNREC_CTX_PER_GPU=4 4 3 "-F -D" "-E -D" > $O/ab_order_ctx3_vs_ctx4.txt 2>&1 || exit 1
cat $O/ab_order_ctx3_vs_ctx4.txt
python3 tools/ab_cli_list.py env:PHNREC_NO_OVERLAP=1 env:PHNREC_NO_OVERLAP=1,PHNREC_NO_ORDER=1 4 3 "-F" "-F -D" "-E -D" > $O/ab_order_vs_neither.txt 2>&1 || exit 1
cat $O/ab_order_vs_neither.txt
PHNREC_NO_OVERLAP=1 TIMELINE_REPS=4 python3 tools/cli_timeline.py 10000 -F -D > $O/timeline_x4.txt 2>&1
PHNREC_NO_OVERLAP=1 TIMELINE_REPS=4 python3 tools/cli_timeline.py 10000 -E -D >> $O/timeline_x4.txt 2>&1
cat $O/timeline_x4.txt
