#!/bin/bash
# Divergence audit of one k_expand instantiation: VGPR/SGPR line, then what LLVM's uniformity analysis says
# about the optimised IR -- loops with a divergent exit (every value carried round such a loop is a VGPR) and
# the ROOT divergent values (divergent although none of their operands is: joins of lane-predicated branches).
# usage: tools/uniformity.sh "1,true,false,true" [-DFLAG ...]
set -e
cd "$(dirname "$0")/.."
python tools/kernel_resources.py --one=$1 "${@:2}" 2>&1 | grep -E "k_expand|error"
mkdir -p /tmp/st && cd /tmp/st
/opt/rocm/bin/hipcc -std=c++17 -O3 --offload-arch=gfx950 -I"$OLDPWD/include" -I"$OLDPWD/stcsp-solver_amd/csrc" "${@:2}" --cuda-device-only -S -emit-llvm \
    -fno-discard-value-names /tmp/engine_one.hip -o one.ll 2>/dev/null
/opt/rocm/lib/llvm/bin/opt -passes='print<uniformity>' -disable-output one.ll 2> uni.txt
awk '/UniformityInfo for function .*k_expand/{f=1} f' uni.txt > uni_k.txt
sed -n '/CYCLES WITH DIVERGENT EXIT/,/^$/p' uni_k.txt | cut -c1-160
python3 - <<'PY'
import re
lines = open('/tmp/st/uni_k.txt').read().splitlines()
div, defs, cur = set(), {}, None
for i, l in enumerate(lines):
    m = re.match(r'BLOCK (\S+)', l)
    if m: cur = m.group(1)
    m = re.match(r'\s*(DIVERGENT:)?\s*(%[\w\.]+) = (.*)', l)
    if m:
        defs[m.group(2)] = (i, cur, m.group(3))
        if m.group(1): div.add(m.group(2))
print("root divergent values:")
for v in sorted(div, key=lambda v: defs[v][0]):
    i, b, rhs = defs[v]
    if not (set(re.findall(r'%[\w\.]+', rhs)) & div) and not re.match(r'(load|atomicrmw|cmpxchg|tail call|call)', rhs):
        print("  ", b[:60], v, rhs[:150])
print("divergent branches:", sum('DIVERGENT:' in l and ' br i1' in l for l in lines), " divergent phis of env/hd:",
      sum(bool(re.search(r'DIVERGENT:.*(env|hd)\.sroa.*phi', l)) for l in lines))
PY
