#!/bin/bash
# GPU session 6 of round 5: posterior kernels in queueing order (lcrc_set_launch_order) with and without the overlapped decoder
set -u
O=gpurun_out/s6
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 300 python -m pytest tests/test_gpu_frontend.py tests/test_gpu_parity.py -m gpu -x -q -k "decoder or reserve or fail" > $O/tests_decoder.log 2>&1
rc=$?; echo "decoder tests rc=$rc"; tail -3 $O/tests_decoder.log
[ $rc -eq 0 ] || exit $rc
# A = neither (round 4's arrangement), B = both
python3 tools/ab_cli_list.py env:PHNREC_NO_ORDER=1,PHNREC_NO_OVERLAP=1 - 4 3 "" "-E" "-F" "-F -D" "-E -D" > $O/ab_none_vs_both.txt 2>&1 || exit 1
cat $O/ab_none_vs_both.txt
# A = order only, B = overlap only
python3 tools/ab_cli_list.py env:PHNREC_NO_OVERLAP=1 env:PHNREC_NO_ORDER=1 4 3 "-F -D" "-E -D" > $O/ab_order_vs_overlap.txt 2>&1 || exit 1
cat $O/ab_order_vs_overlap.txt
# two contexts per GPU: neither against both
python3 tools/ab_cli_list.py env:PHNREC_NO_ORDER=1,PHNREC_NO_OVERLAP=1,PHNREC_CTX_PER_GPU=2 env:PHNREC_CTX_PER_GPU=2 4 2 "-F" "-F -D" "-E -D" > $O/ab_ctx2.txt 2>&1 || exit 1
cat $O/ab_ctx2.txt
TIMELINE_REPS=4 python3 tools/cli_timeline.py 10000 -F > $O/timeline_x4.txt 2>&1
TIMELINE_REPS=4 python3 tools/cli_timeline.py 10000 -F -D >> $O/timeline_x4.txt 2>&1
TIMELINE_REPS=4 python3 tools/cli_timeline.py 10000 -E -D >> $O/timeline_x4.txt 2>&1
cat $O/timeline_x4.txt
