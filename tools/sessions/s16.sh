#!/bin/bash
# GPU session 16 of round 5: lists take the GPU front-end by themselves on one GPU too: every GPU test, the driver's bench form
set -u
O=gpurun_out/s16
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -3 $O/gputests.log
[ $rc -eq 0 ] || exit $rc
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
python3 tools/bench_summary.py $O/bench_driver.json > $O/bench_summary.txt 2>&1
tail -24 $O/bench_summary.txt
timeout -k 10 400 python tools/fuzz_cli.py 11 12 > $O/fuzz.txt 2>&1; tail -2 $O/fuzz.txt
