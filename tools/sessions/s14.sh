#!/bin/bash
# GPU session 14 of round 5 (experiment): decoder kernel confined to n CUs by a CU-masked stream
set -u
O=gpurun_out/s14
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
for n in 16 64; do
  python3 tools/ab_cli_list.py - env:LCRC_DEC_CUS=$n 4 3 "-F -D" > $O/ab_cus$n.txt 2>&1 || { tail -3 $O/ab_cus$n.txt; exit 1; }
  grep median $O/ab_cus$n.txt
done
LCRC_DEC_CUS=16 TIMELINE_REPS=4 python3 tools/cli_timeline.py 10000 -F -D > $O/timeline_cus16.txt 2>&1
grep -A12 "^files" $O/timeline_cus16.txt
