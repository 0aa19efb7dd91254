#!/usr/bin/env python3
"""Compile engine.hip for gfx950 with -Rpass-analysis=kernel-resource-usage and print one line per
kernel: VGPRs, SGPRs, spills, scratch, occupancy.
Usage: tools/kernel_resources.py [pattern] [-DFLAG ...]
       tools/kernel_resources.py --one=1,true,false,true [-DFLAG ...]   (only that k_expand instantiation: seconds)"""
import re
import subprocess
import sys
from pathlib import Path

CSRC = Path(__file__).resolve().parent.parent / "stcsp-solver_amd" / "csrc"
pat = next((a for a in sys.argv[1:] if not a.startswith("-")), "")
extra = [a for a in sys.argv[1:] if a.startswith("-") and not a.startswith("--one=")]
one = next((a[6:] for a in sys.argv[1:] if a.startswith("--one=")), None)
src = CSRC / "engine.hip"
if one:
    src = Path("/tmp/engine_one.hip")
    src.write_text('#include <hip/hip_runtime.h>\n#include <climits>\n#include "device_types.hpp"\n#include "stcsp_engine.h"\nusing namespace stcsp;\n'
                   f'#include "dev_kernels.hpp"\ntemplate __global__ void stcsp::dev::k_expand<{one}>(const stcsp::dev::Ctx *, const stcsp::dev::Plan *, unsigned, uint32_t);\n')
cmd = ["/opt/rocm/bin/hipcc", "-std=c++17", "-O3", "-fPIC", "--offload-arch=gfx950", f"-I{CSRC.parent.parent / 'include'}",
       f"-I{CSRC}", "-Rpass-analysis=kernel-resource-usage", "-c", str(src), "-o", "/tmp/engine_res.o"] + extra
if "-save-temps" in extra:
    import os
    os.makedirs("/tmp/st", exist_ok=True)
    os.chdir("/tmp/st")
r = subprocess.run(cmd, capture_output=True, text=True)
if r.returncode != 0:
    print(r.stderr[-4000:])
    sys.exit(1)
cur = None
rows = {}
for line in r.stderr.splitlines():
    m = re.search(r"remark: +(.*?): (.*?) \[-Rpass", line) or re.search(r"remark: +Function Name: (.*?) \[-Rpass", line)
    if "Function Name:" in line:
        cur = re.search(r"Function Name: (\S+)", line).group(1)
        rows[cur] = {}
    elif m and cur:
        rows[cur][m.group(1).strip()] = m.group(2).strip()
demangle = subprocess.run(["c++filt"], input="\n".join(rows), capture_output=True, text=True).stdout.splitlines()
for name, d in zip(demangle, rows.values()):
    if pat and pat not in name:
        continue
    short = re.sub(r"stcsp::dev::|\(.*", "", name).replace("void ", "")
    print(f"{short:42s} VGPR {d.get('VGPRs','?'):>4s} SGPR {d.get('TotalSGPRs','?'):>4s} spillS {d.get('SGPRs Spill','?'):>4s} "
          f"spillV {d.get('VGPRs Spill','?'):>3s} scratch {d.get('ScratchSize [bytes/lane]','?'):>4s} occ {d.get('Occupancy [waves/SIMD]','?')}")
