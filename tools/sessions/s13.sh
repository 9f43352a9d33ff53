#!/bin/bash
# GPU session 13 of round 5: who slows the posterior kernel in a list -- the same list as parameter files (no front-end kernels)
# with and without the device decoder, beside the waveform modes
set -u
O=gpurun_out/s13
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
TIMELINE_PAR=1 TIMELINE_REPS=4 python3 tools/cli_timeline.py 10000 > $O/timeline_par.txt 2>&1
TIMELINE_PAR=1 TIMELINE_REPS=4 python3 tools/cli_timeline.py 10000 -D >> $O/timeline_par.txt 2>&1
TIMELINE_PAR=1 TIMELINE_REPS=4 PHNREC_LAUNCH_ORDER=1 python3 tools/cli_timeline.py 10000 >> $O/timeline_par.txt 2>&1
TIMELINE_REPS=4 python3 tools/cli_timeline.py 10000 -F -D >> $O/timeline_par.txt 2>&1
cat $O/timeline_par.txt
