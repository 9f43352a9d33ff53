#!/bin/bash
# GPU session 10 of round 5: decoder kernel with the DPP-operand maxima and the writelane history push: tests, A/B on the 4x list
set -u
O=gpurun_out/s10
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 400 python -m pytest tests/test_gpu_frontend.py tests/test_gpu_parity.py -m gpu -x -q -k "decoder or reserve" > $O/tests_decoder.log 2>&1
rc=$?; echo "decoder tests rc=$rc"; tail -3 $O/tests_decoder.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python -m pytest tests/test_gpu_cli.py -m gpu -x -q -k "decoder or four or gpu_front_end or golden or scale" > $O/tests_cli.log 2>&1
rc=$?; echo "cli tests rc=$rc"; tail -3 $O/tests_cli.log
[ $rc -eq 0 ] || exit $rc
python3 tools/ab_cli_list.py phnrec_amd/lib/ab/libr5dec0.so - 4 4 "-F" "-F -D" "-E -D" > $O/ab_dec.txt 2>&1 || exit 1
grep median $O/ab_dec.txt
TIMELINE_REPS=4 python3 tools/cli_timeline.py 10000 -F -D > $O/timeline_x4.txt 2>&1
grep -A8 "^files" $O/timeline_x4.txt
