#!/bin/bash
# GPU session 1 of round 5: helper-wave A/B, stamps, EN trace, GPU tests, the driver's bench form
set -u
O=gpurun_out/s1
mkdir -p $O
python tools/ab_kernel.py phnrec_amd/lib/ab/libnohelp.so phnrec_amd/lib/ab/libhelp.so 4096 > $O/ab4096.txt 2>&1 || exit 1
python tools/ab_kernel.py phnrec_amd/lib/ab/libnohelp.so phnrec_amd/lib/ab/libhelp.so 2560 > $O/ab2560.txt 2>&1 || exit 1
python tools/ab_kernel.py phnrec_amd/lib/ab/libr5base.so phnrec_amd/lib/ab/libnohelp.so 8192 > $O/ab8192_refactor.txt 2>&1 || exit 1
python tools/ab_kernel.py phnrec_amd/lib/ab/libr5base.so phnrec_amd/lib/ab/libnohelp.so 4096 > $O/ab4096_refactor.txt 2>&1 || exit 1
echo "ab done"
LCRC_BM=16 python tools/stamp_profile.py PHN_EN_TIMIT_LCRC_N500 4096 > $O/stamps_en4096_help.txt 2>&1 || exit 1
LCRC_DBG=16 LCRC_BM=16 python tools/stamp_profile.py PHN_EN_TIMIT_LCRC_N500 4096 > $O/stamps_en4096_nohelp.txt 2>&1 || exit 1
python tools/en_repro.py > $O/en_repro.txt 2>&1 || exit 1
echo "traces done"
timeout -k 10 500 python -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1
rc=$?
echo "tests rc=$rc"
tail -3 $O/gputests.log
[ $rc -eq 0 ] || exit $rc
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench.err
echo "bench rc=$?"
