#!/bin/bash
# GPU session 4 of round 5 (library back to round 4's kernels): contexts per GPU with the decoder on the device, the decoder
# kernel at issue priority 3 (A/B library), timelines of -F -D at 3 and 4 contexts
set -u
O=gpurun_out/s4
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
python3 tools/ab_cli_list.py - env:PHNREC_CTX_PER_GPU=4 4 3 "-F" "-F -D" "-E -D" > $O/ab_ctx4.txt 2>&1 || exit 1
cat $O/ab_ctx4.txt
python3 tools/ab_cli_list.py - env:PHNREC_CTX_PER_GPU=5 4 2 "-F -D" "-E -D" > $O/ab_ctx5.txt 2>&1 || exit 1
cat $O/ab_ctx5.txt
python3 tools/ab_cli_list.py - env:PHNREC_CTX_PER_GPU=2 4 2 "-F" "-F -D" "-E -D" > $O/ab_ctx2.txt 2>&1 || exit 1
cat $O/ab_ctx2.txt
python3 tools/ab_cli_list.py - phnrec_amd/lib/ab/libdecprio.so 4 3 "-F -D" "-E -D" > $O/ab_decprio.txt 2>&1 || exit 1
cat $O/ab_decprio.txt
TIMELINE_REPS=4 python3 tools/cli_timeline.py 10000 -F -D > $O/timeline_x4.txt 2>&1
PHNREC_CTX_PER_GPU=4 TIMELINE_REPS=4 python3 tools/cli_timeline.py 10000 -F -D >> $O/timeline_x4.txt 2>&1
PHNREC_CTX_PER_GPU=2 TIMELINE_REPS=4 python3 tools/cli_timeline.py 10000 -F -D >> $O/timeline_x4.txt 2>&1
cat $O/timeline_x4.txt
