set -u
timeout -k 10 900 python -m pytest tests -m gpu -q --durations=8 > gpurun_out/r03_t3.log 2>&1; rc=$?; echo pytest rc=$rc; tail -12 gpurun_out/r03_t3.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
timeout -k 10 400 python bench.py > gpurun_out/r03_b3.json 2> gpurun_out/r03_b3.err; echo bench rc=$?
python tools/traps_bench.py > gpurun_out/r03_traps_bench.txt 2>&1; cat gpurun_out/r03_traps_bench.txt
