#!/bin/bash
# bash tools/build_variant.sh NAME "<extra hipcc flags>": build grafx_amd/lib/NAME.so with extra flags (objects in their
# own directory); the live libgrafx_amd.so is not touched.  Use with tools/ab.sh.
R=${GRAFT_REPO_ROOT:-$(pwd)}
GRAFX_AMD_LIB=$R/grafx_amd/lib/$1.so GRAFX_HIPCC_FLAGS="$2" python -m grafx_amd.build --force > /tmp/build_$1.log 2>&1 \
  || { grep -E "error|Error" -A3 /tmp/build_$1.log | head -30; exit 1; }
echo "built grafx_amd/lib/$1.so"
