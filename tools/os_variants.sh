#!/bin/bash
# one-shot dynamics ablations (timing only): build the variants here (no GPU needed), time them on the box with
#   bash tools/ab.sh "OS0 OSSPEC OSNOXCD OSNOMATH OSNOHIST OSCOPY" 1 python tools/microbench.py comp1 --rows 8192
for v in "OS0:" "OSSPEC:-DGFX_OS_SPEC" "OSNOXCD:-DGFX_OS_NOXCD" "OSNOMATH:-DGFX_OS_NOMATH" "OSNOHIST:-DGFX_OS_NOHIST" "OSCOPY:-DGFX_OS_COPY"; do
  bash tools/build_variant.sh "${v%%:*}" "${v#*:}" &
done
wait
