set -u
cd tools/ubench
TIMEFORMAT='wall %R s'
for i in 1 2 3; do { time ./hip_startup > ../../gpurun_out/hs_$i.txt; } 2>> ../../gpurun_out/hs_time.txt; done
for i in 1 2 3; do { time FAST_EXIT=1 ./hip_startup > /dev/null; } 2>> ../../gpurun_out/hs_time.txt; done
for v in HSA_ENABLE_SDMA=0 HSA_ENABLE_INTERRUPT=0 HIP_FORCE_DEV_KERNARG=1 ROCR_VISIBLE_DEVICES=0 GPU_MAX_HW_QUEUES=1 HIP_LAUNCH_BLOCKING=0; do echo "== $v" >> ../../gpurun_out/hs_env.txt; env $v FAST_EXIT=1 ./hip_startup 2>&1 | grep -E "hipInit|hipFree|StreamCreate|total" >> ../../gpurun_out/hs_env.txt; done
cd ../..
M=tests/golden/models/PHN_CZ_SPDAT_LCRC_N1500
{ for i in 1 2 3; do { time phnrec_amd/bin/phnrec > /dev/null; } 2>&1 | sed 's/^/help: /'; done
for i in 1 2 3; do { time phnrec_amd/bin/phnrec -c $M -s post -i tests/golden/PHN_CZ_SPDAT_LCRC_N1500/test.lop -o /tmp/x.rec; } 2>&1 | sed 's/^/post->str no GPU: /'; done
for i in 1 2 3 4 5; do { time PHNREC_STATS=1 LCRC_TRACE_STARTUP=1 phnrec_amd/bin/phnrec -c $M -i tests/golden/test.raw -o /tmp/y.rec; } 2>&1 | sed 's/^/wf->str: /'; done; } > gpurun_out/single_file_probe.txt 2>&1
for n in 2 3 4; do echo "== ctx per gpu $n"; PHNREC_CTX_PER_GPU=$n python -c "
import bench, json
r = bench.sharded_list_leg(1, [0], 10000)
print(json.dumps({k: r[k] for k in ('host_frontend','gpu_frontend_F','gpu_frontend_decoder_F_D','host_ceiling')}))
"; done > gpurun_out/ctx_sweep.txt 2>&1
for g in 2 4 8; do echo "== -g $g logical on one GPU"; python -c "
import bench, json
r = bench.sharded_list_leg($g, [0]*$g, 10000)
print(json.dumps({k: r[k] for k in ('host_frontend','gpu_frontend_F','gpu_frontend_decoder_F_D')}))
"; done >> gpurun_out/ctx_sweep.txt 2>&1
timeout -k 10 900 python -m pytest tests -m gpu -q --durations=12 > gpurun_out/r03_t2.log 2>&1; echo pytest rc=$?; tail -30 gpurun_out/r03_t2.log
