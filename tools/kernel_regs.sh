#!/bin/bash
# VGPR / SGPR / scratch of the kernels as compiled with the given extra flags:  tools/kernel_regs.sh [-DFLAG=1 ...]
R=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -ffp-contract=off -std=c++17 -I$R/include -I$R/jurassic-gpu_amd/csrc \
  -DJUR_ND=${JUR_ND:-100} -DJUR_NG=${JUR_NG:-30} "$@" -S --cuda-device-only -o $T/k.s $R/jurassic-gpu_amd/csrc/jur_kernels.hip 2>/dev/null
python3 - $T/k.s <<'PY'
import re, sys
t = open(sys.argv[1]).read()
m = t[t.index("amdhsa.kernels:"):]
for b in m.split("  - .agpr_count:")[1:]:
    if "jur_" not in b.split(".name:")[1].split()[0]: continue
    n = re.search(r"\.name:\s+(\S+)", b).group(1)
    g = lambda k: re.search(r"\.%s:\s+(\d+)" % k, b).group(1)
    short = re.sub(r"^_ZN\d+_GLOBAL__N_1\d+", "", n)[:40]
    print("%-42s vgpr %3s sgpr %3s scratch %s" % (short, g("vgpr_count"), g("sgpr_count"), g("private_segment_fixed_size")))
PY
rm -rf $T
