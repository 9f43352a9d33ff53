#!/bin/bash
# BASELINE configs[4]: all four LCRC systems at once, two GPUs each, on one 8-GPU node.
#   tools/run_four_systems.sh CZ_DIR CZ.scp HU_DIR HU.scp RU_DIR RU.scp EN_DIR EN.scp [extra phnrec flags, e.g. -F -D]
# Each system is one `phnrec -g 2` process pinned to its GPU pair with HIP_VISIBLE_DEVICES (the processes share
# nothing: every GPU holds its system's weights, utterances never cross GPUs); MLFs go next to the lists.
# PHNREC_GPU_PAIRS="0,1 2,3 4,5 6,7" (default) names the pairs; a pair may repeat a GPU ("0,0": both logical
# GPUs of that system on one device, PHNREC_DEVICE_MAP) so that a smaller node -- or a 1-GPU box, with
# "0,0 0,0 0,0 0,0" -- runs the same four-process arrangement.
set -eu
BIN="${PHNREC_BIN:-$(dirname "$0")/../phnrec_amd/bin/phnrec}"
[ $# -ge 8 ] || { sed -n 2,8p "$0"; exit 1; }
dirs=("$1" "$3" "$5" "$7"); lists=("$2" "$4" "$6" "$8"); shift 8
read -r -a pairs <<< "${PHNREC_GPU_PAIRS:-0,1 2,3 4,5 6,7}"
[ ${#pairs[@]} -eq 4 ] || { echo "PHNREC_GPU_PAIRS needs four pairs" >&2; exit 1; }
pids=()
for i in 0 1 2 3; do
    a="${pairs[$i]%,*}"; b="${pairs[$i]#*,}"
    if [ "$a" = "$b" ]; then vis="$a"; map="0,0"; else vis="$a,$b"; map="0,1"; fi
    HIP_VISIBLE_DEVICES="$vis" PHNREC_DEVICE_MAP="$map" PHNREC_STATS=1 \
        "$BIN" -c "${dirs[$i]}" -l "${lists[$i]}" -m "${lists[$i]%.*}.mlf" -g 2 "$@" &
    pids+=($!)
done
rc=0
for p in "${pids[@]}"; do wait "$p" || rc=$?; done
exit $rc
