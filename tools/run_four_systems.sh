#!/bin/bash
# BASELINE configs[4]: all four LCRC systems at once, two GPUs each, on one 8-GPU node.
#   tools/run_four_systems.sh CZ_DIR CZ.scp HU_DIR HU.scp RU_DIR RU.scp EN_DIR EN.scp [extra phnrec flags, e.g. -F -D]
# Each system is one `phnrec -g 2` process pinned to its GPU pair with HIP_VISIBLE_DEVICES (the processes share
# nothing: every GPU holds its system's weights, utterances never cross GPUs); MLFs go next to the lists.
set -eu
BIN="$(dirname "$0")/../phnrec_amd/bin/phnrec"
[ $# -ge 8 ] || { sed -n 2,5p "$0"; exit 1; }
dirs=("$1" "$3" "$5" "$7"); lists=("$2" "$4" "$6" "$8"); shift 8
pids=()
for i in 0 1 2 3; do
    HIP_VISIBLE_DEVICES=$((2 * i)),$((2 * i + 1)) PHNREC_STATS=1 \
        "$BIN" -c "${dirs[$i]}" -l "${lists[$i]}" -m "${lists[$i]%.*}.mlf" -g 2 "$@" &
    pids+=($!)
done
rc=0
for p in "${pids[@]}"; do wait "$p" || rc=$?; done
exit $rc
