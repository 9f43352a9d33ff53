#!/usr/bin/env python3
"""Register / spill / scratch table of every kernel of a .hip file (hipcc -Rpass-analysis=kernel-resource-usage).

    python tools/resource_table.py [file.hip ...]        (default: the four kernel files of phnrec_amd/csrc)
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "phnrec_amd", "csrc")
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-mllvm", "-amdgpu-mfma-vgpr-form",
         "-Rpass-analysis=kernel-resource-usage", "-c", "-o", "/dev/null"]


def table(path):
    err = subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + [path], cwd=CSRC, capture_output=True, text=True).stderr
    rows, cur = [], None
    for line in err.splitlines():
        m = re.search(r"remark:\s+(?:Function )?Name: (\S+)", line)
        if m:
            cur = {"name": subprocess.run(["c++filt", m.group(1)], capture_output=True,
                                          text=True).stdout.strip()}
            rows.append(cur)
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
        if m and cur is not None:
            cur[m.group(1).strip()] = int(m.group(2))
    return rows


def main():
    files = sys.argv[1:] or ["lcrc_kernels.hip", "traps_kernels.hip", "phndec_kernels.hip", "frontend_kernels.hip"]
    print("%-88s %5s %5s %6s %6s %7s %4s" % ("kernel", "VGPR", "SGPR", "vspill", "sspill", "scratch", "occ"))
    for f in files:
        for r in table(f):
            name = re.sub(r"^void phnrec::", "", r["name"]).replace("(phnrec::LcrcParams)", "")
            print("%-88s %5d %5d %6d %6d %7d %4d" % (name[:88], r.get("VGPRs", -1), r.get("TotalSGPRs", -1),
                                                     r.get("VGPRs Spill", -1), r.get("SGPRs Spill", -1),
                                                     r.get("ScratchSize", -1), r.get("Occupancy", -1)))


if __name__ == "__main__":
    main()
