#!/bin/bash
# Copies the summaries of a tools/refresh_profiles.sh run (gpurun_out/refresh_<tag>/) into profiles/ under the round's names.
#   tools/collect_profiles.sh r02
set -e
TAG=${1:-r06}
R=gpurun_out/refresh_$TAG
P=profiles
clean() { grep -v amdgpu.ids "$1" > "$2"; }
first() { ls $1 | head -1; }     # (a traced run may leave one file per process)
cp $R/bench.json $P/${TAG}_bench.json
cp $R/bench_driver_form.json $P/${TAG}_bench_driver_form.json
for f in bench_line bench_driver_form_line; do [ -f $R/$f.json ] && cp $R/$f.json $P/${TAG}_$f.json; done      # what stdout carried
cp $(first "$R/stats/*/*kernel_stats.csv") $P/${TAG}_kernel_stats_all_launches.csv
[ -f $R/steady_kernel_stats.csv ] && cp $R/steady_kernel_stats.csv $P/${TAG}_kernel_stats.csv      # the timed launches only
[ -f $R/bench_profiled.json ] && cp $R/bench_profiled.json $P/${TAG}_bench_profiled.json              # the traced run's own line
[ -f $R/bench_2ranks_one_gpu.json ] && cp $R/bench_2ranks_one_gpu.json $P/${TAG}_bench_2ranks_one_gpu.json
[ -f $R/system_sweep_fused_only.txt ] && clean $R/system_sweep_fused_only.txt $P/${TAG}_system_sweep_fused_only.txt
[ -f $R/startup_probe.txt ] && clean $R/startup_probe.txt $P/${TAG}_startup_probe.txt
[ -f $R/cli_contexts.txt ] && clean $R/cli_contexts.txt $P/${TAG}_cli_contexts.txt
for f in cli_timeline cli_sweep pinned_read pipeline_trace host_mem_load selftest_gpu_ln; do [ -f $R/$f.txt ] && clean $R/$f.txt $P/${TAG}_$f.txt; done
cp $R/kernel_trace_head.csv $P/${TAG}_kernel_trace_head.csv
python3 tools/pmc_summary.py $TAG $R > /dev/null
clean $R/phase_stamps.txt $P/${TAG}_phase_stamps.txt
clean $R/system_sweep.txt $P/${TAG}_system_sweep.txt
clean $R/small_launch_sweep.txt $P/${TAG}_small_launch_sweep.txt
clean $R/traps_bench.txt $P/${TAG}_traps_bench.txt
cp $(first "$R/traps_stats/*/*kernel_stats.csv") $P/${TAG}_traps_kernel_stats.csv
cp $(first "$R/wave_stats/*/*kernel_stats.csv") $P/${TAG}_waveform_entry_kernel_stats.csv
[ -f $R/cli_throughput.txt ] && clean $R/cli_throughput.txt $P/${TAG}_cli_throughput.txt
[ -f $R/ubench.txt ] && clean $R/ubench.txt $P/${TAG}_ubench.txt
clean $R/split_f16_bench.txt $P/${TAG}_split_f16_bench.txt
[ -f $R/split_phase_stamps.txt ] && clean $R/split_phase_stamps.txt $P/${TAG}_split_f16_phase_stamps.txt
cp $(first "$R/split_stats/*/*kernel_stats.csv") $P/${TAG}_split_f16_kernel_stats.csv
[ -f $R/split_ubench.txt ] && cp $R/split_ubench.txt $P/${TAG}_split_f16_ubench.txt
python3 - "$R" "$P/${TAG}_split_f16_pmc.json" <<'PY'
import collections, csv, glob, json, sys
agg = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/split_pmc/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "lcrc_fused_kernel<42, 69, 9, 4, true, 2, false, false, 1>" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: {"mean": sum(v) / len(v), "launches": len(v)} for k, v in sorted(agg.items())}
d = {}
if "SQ_VALU_MFMA_BUSY_CYCLES" in out and "SQ_INSTS_MFMA" in out:
    d["mfma_cycles_per_instruction"] = out["SQ_VALU_MFMA_BUSY_CYCLES"]["mean"] / out["SQ_INSTS_MFMA"]["mean"]
    d["mfma_busy_cycles_per_simd"] = out["SQ_VALU_MFMA_BUSY_CYCLES"]["mean"] / 1024.0
if "SQ_INSTS_VMEM_RD" in out:
    d["vmem_read_instructions_per_wave"] = out["SQ_INSTS_VMEM_RD"]["mean"] / 1024.0
if out:        # (only when the split-f16 PMC pass was part of the refresh run)
    json.dump({"kernel": "lcrc_fused_kernel<..., ARITH = 1> (CZ, 8192 frames, split-f16)", "counters": out, "derived": d},
              open(sys.argv[2], "w"), indent=1)
print(json.dumps(d))
PY
