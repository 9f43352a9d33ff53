#!/bin/bash
# GPU session 3a of round 5: GPU tests, the driver's bench form (new legs: -E -D, weak list at N = 1, as_g8_default, four_systems)
set -u
O=gpurun_out/s3
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 700 python -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1
rc=$?
echo "tests rc=$rc"; tail -3 $O/gputests.log
[ $rc -eq 0 ] || exit $rc
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
echo "bench done"
python3 tools/bench_summary.py $O/bench_driver.json > $O/bench_summary.txt 2>&1
cat $O/bench_summary.txt
