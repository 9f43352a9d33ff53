set -u
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -x -k "not fuzzed" > gpurun_out/r03_t5.log 2>&1; rc=$?; echo rc=$rc; tail -4 gpurun_out/r03_t5.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
python tools/system_sweep.py 2048 3072 4096 4100 5120 6144 8192 10240 12288 16384 32768 > gpurun_out/r03_sweep_auto.txt 2>&1
SWEEP_NO_SPLIT=1 python tools/system_sweep.py 4100 6144 12288 > gpurun_out/r03_sweep_nosplit.txt 2>&1
cat gpurun_out/r03_sweep_auto.txt gpurun_out/r03_sweep_nosplit.txt
python tools/small_launch_sweep.py PHN_CZ_SPDAT_LCRC_N1500 > gpurun_out/r03_small_sweep.txt 2>&1; tail -40 gpurun_out/r03_small_sweep.txt
