#!/bin/bash
# GPU session 8 of round 5: the policy as set (order with -D, overlap at two contexts): every GPU test, launch sizes under the
# order, -F in order with 65 536-frame launches
set -u
O=gpurun_out/s8
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 700 python -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -3 $O/gputests.log
[ $rc -eq 0 ] || exit $rc
python3 tools/ab_cli_list.py - - 4 3 "-F" "-F -b 65536" "-F -D" "-F -D -b 32768" "-F -D -b 131072" "-E -D" "-E -D -b 32768" "-E -D -b 131072" > $O/ab_batch.txt 2>&1 || exit 1
grep median $O/ab_batch.txt
python3 tools/ab_cli_list.py - env:PHNREC_LAUNCH_ORDER=1 4 3 "-F -b 65536" "-E -b 65536" "" > $O/ab_order_nonD_b64k.txt 2>&1 || exit 1
grep median $O/ab_order_nonD_b64k.txt
