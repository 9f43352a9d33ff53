#!/bin/bash
# GPU session 11 of round 5: every GPU test and the driver's bench form after the source split (no code changes)
set -u
O=gpurun_out/s11
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -3 $O/gputests.log
[ $rc -eq 0 ] || exit $rc
python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
python3 tools/bench_summary.py $O/bench_driver.json > $O/bench_summary.txt 2>&1
cat $O/bench_summary.txt
