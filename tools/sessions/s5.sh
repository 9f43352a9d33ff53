#!/bin/bash
# GPU session 5 of round 5: decoder overlapped with the next launch (lcrc_set_decoder_overlap): tests, then A/B against the
# library and CLI before it (phnrec_amd/lib/ab/r5base: both files swapped), contexts per GPU 2 and 3
set -u
O=gpurun_out/s5
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 300 python -m pytest tests/test_gpu_frontend.py tests/test_gpu_parity.py -m gpu -x -q -k "decoder or reserve" > $O/tests_decoder.log 2>&1
rc=$?; echo "decoder tests rc=$rc"; tail -5 $O/tests_decoder.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python -m pytest tests/test_gpu_cli.py -m gpu -x -q > $O/tests_cli.log 2>&1
rc=$?; echo "cli tests rc=$rc"; tail -5 $O/tests_cli.log
[ $rc -eq 0 ] || exit $rc
python3 tools/ab_cli_list.py - env:PHNREC_CTX_PER_GPU=2 4 3 "-F" "-F -D" "-E -D" > $O/ab_ctx2.txt 2>&1 || exit 1
cat $O/ab_ctx2.txt
TIMELINE_REPS=4 python3 tools/cli_timeline.py 10000 -F -D > $O/timeline_x4.txt 2>&1
TIMELINE_REPS=4 python3 tools/cli_timeline.py 10000 -E -D >> $O/timeline_x4.txt 2>&1
PHNREC_CTX_PER_GPU=2 TIMELINE_REPS=4 python3 tools/cli_timeline.py 10000 -E -D >> $O/timeline_x4.txt 2>&1
cat $O/timeline_x4.txt
