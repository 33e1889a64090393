#!/bin/bash
# what a non-root process on the GPU box can read about the shader clock (for bench.py's `sustained.sclk_mhz_mean`)
for d in /sys/class/drm/card*/device; do
  echo "== $d"; ls $d | grep -i -E "clk|freq|power|hwmon" | head -20
  for f in pp_dpm_sclk pp_dpm_mclk; do [ -r $d/$f ] && { echo "-- $f"; cat $d/$f; }; done
  for h in $d/hwmon/hwmon*; do echo "-- $h"; ls $h | head -40; for f in freq1_input freq1_label power1_average power1_input power1_cap; do [ -r $h/$f ] && echo "$f=$(cat $h/$f)"; done; done
done
python3 -c "import amdsmi; print('amdsmi ok', amdsmi.__file__)" 2>&1 | tail -1
which rocm-smi amd-smi; timeout 20 rocm-smi --showclocks 2>&1 | head -20
timeout 20 rocm-smi --showpower 2>&1 | head -12
