#!/bin/bash
# Register / scratch / occupancy report of every kernel of one HIP source (compile only; no GPU needed):
#   tools/kernel_regs.sh pylc_amd/csrc/conv_pl.hip [filter]
# Prints: kernel | VGPRs | AGPRs | scratch bytes per lane | occupancy (waves per SIMD) | LDS bytes (static)
set -o pipefail
src=$1; filt=${2:-.}
cd "$(dirname "$src")" || exit 1
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize \
    -Rpass-analysis=kernel-resource-usage -c "$(basename "$src")" -o /dev/null 2>&1 |
  awk '/Function Name:/ {name=$0; sub(/.*Function Name: /,"",name); sub(/ \[-Rpass.*/,"",name)}
       / VGPRs:/ {v=$0; sub(/.* VGPRs: /,"",v); sub(/ \[.*/,"",v)}
       / AGPRs:/ {a=$0; sub(/.* AGPRs: /,"",a); sub(/ \[.*/,"",a)}
       /ScratchSize/ {s=$0; sub(/.*: /,"",s); sub(/ \[.*/,"",s)}
       /Occupancy/ {o=$0; sub(/.*: /,"",o); sub(/ \[.*/,"",o)}
       /LDS Size/ {l=$0; sub(/.*: /,"",l); sub(/ \[.*/,"",l); print name, "| vgpr", v, "| agpr", a, "| scratch", s, "| occ", o, "| lds", l}' |
  c++filt | grep -E "$filt"
