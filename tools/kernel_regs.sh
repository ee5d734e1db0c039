#!/bin/bash
# VGPRs / spills / scratch / occupancy / LDS of the kernels of one translation unit, from the compiler's own resource remarks
# (no GPU needed):   tools/kernel_regs.sh nrs_inst_f32_muller.hip 'k_density_tiled|k_forces_lists' [extra -D flags]
set -e
cd "$(dirname "$0")/../nereus_amd/csrc"
src=$1; pat=${2:-.}; shift; shift || true
out=$(mktemp)
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -w --cuda-device-only \
    -Rpass-analysis=kernel-resource-usage "$@" -c -o /dev/null $src > $out 2>&1 || { cat $out; exit 1; }
python3 - "$out" "$pat" <<'PY'
import re, subprocess, sys
txt = open(sys.argv[1]).read()
for b in re.split(r"remark: [^\n]*Function Name: ", txt)[1:]:
    name = b.split()[0]
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    if not re.search(sys.argv[2], dem):
        continue
    g = lambda k: (re.search(k + r": (\d+)", b) or [0, "?"])[1]
    print("vgpr %3s spill %3s scratch %4s occ %s lds %6s  %s" % (g("VGPRs"), g("VGPRs Spill"), g(r"ScratchSize \[bytes/lane\]"),
          g(r"Occupancy \[waves/SIMD\]"), g(r"LDS Size \[bytes/block\]"), dem[:140]))
PY
rm -f $out
