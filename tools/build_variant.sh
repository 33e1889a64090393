#!/bin/bash
# bash tools/build_variant.sh NAME "<extra hipcc flags>": rebuild the library with extra flags and keep it as grafx_amd/lib/NAME.so
GRAFX_HIPCC_FLAGS="$2" python -m grafx_amd.build --force > /tmp/build_$1.log 2>&1 || { grep -E "error|Error" -A3 /tmp/build_$1.log | head -30; exit 1; }
cp grafx_amd/lib/libgrafx_amd.so grafx_amd/lib/$1.so && echo "built $1"
