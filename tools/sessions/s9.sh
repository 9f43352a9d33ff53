#!/bin/bash
# GPU session 9 of round 5: -F with the host libm's ln() sequence (bit-identical features), auto-selected from two GPUs on
set -u
O=gpurun_out/s9
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -3 $O/gputests.log
[ $rc -eq 0 ] || exit $rc
./phnrec_amd/bin/phnrec --selftest-gpu-ln > $O/selftest_gpu_ln.txt 2>&1; cat $O/selftest_gpu_ln.txt
python3 tools/frontend_bench.py 200 > $O/frontend_bench.txt 2>&1; tail -5 $O/frontend_bench.txt
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
python3 tools/bench_summary.py $O/bench_driver.json > $O/bench_summary.txt 2>&1
cat $O/bench_summary.txt
