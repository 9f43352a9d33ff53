#!/bin/bash
# The list pipeline's host code under ThreadSanitizer ON the GPU box (the HIP runtime itself is not instrumented): contexts
# that come up beside the list, clones waiting for their base, contexts left out, the fault paths.  usage: tsan_gpu_probe.sh
set -u
cd "$(dirname "$0")/.."
T=$(mktemp -d /tmp/tsanprobe.XXXX)
python3 - "$T" <<'PY'
import sys, os, numpy as np
d=sys.argv[1]; rng=np.random.default_rng(5); lines=[]
for i in range(120):
    n=int(rng.integers(1600, 20000)); t=np.arange(n)/8000.0
    x=0.3*32767/5*sum(np.sin(2*np.pi*f*t) for f in (200,700,1300,2100,3400))+rng.normal(0,1000,n)
    p=os.path.join(d,"f%03d.raw"%i); np.clip(x,-32768,32767).astype("<i2").tofile(p); lines.append(p)
open(os.path.join(d,"l.scp"),"w").write("\n".join(lines)+"\n")
PY
M=tests/golden/models/PHN_HU_SPDAT_LCRC_N1500
export TSAN_OPTIONS="halt_on_error=0 report_signal_unsafe=0 second_deadlock_stack=1"
for flags in "" "-F" "-F -D"; do
  for g in 1 4; do
    map=$(python3 -c "print(','.join(['0']*$g))")
    echo "== -g $g $flags"
    PHNREC_DEVICE_MAP=$map timeout -k 5 120 setarch x86_64 -R phnrec_amd/bin/phnrec_tsan -c $M -l $T/l.scp -m $T/o_$g.mlf -g $g -b 600 $flags 2> $T/err.txt; echo "rc=$?"
    grep -c "WARNING: ThreadSanitizer" $T/err.txt
    grep -A12 "WARNING: ThreadSanitizer" $T/err.txt | grep -E "WARNING|#[0-3] " | grep -v "libamdhip64\|libhsa" | head -24
  done
done
echo "== launch fault, -g 4 -F -D"
LCRC_FAULT_INJECTION=1 PHNREC_FAIL_LAUNCH_NTH=7 PHNREC_DEVICE_MAP=0,0,0,0 timeout -k 5 120 setarch x86_64 -R phnrec_amd/bin/phnrec_tsan -c $M -l $T/l.scp -m $T/f.mlf -g 4 -b 600 -F -D 2> $T/err.txt; echo "rc=$?"
grep -c "WARNING: ThreadSanitizer" $T/err.txt; tail -2 $T/err.txt
cmp $T/o_1.mlf $T/o_4.mlf && echo "MLFs equal"
rm -rf $T
