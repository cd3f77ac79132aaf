#!/bin/bash
# here (no GPU): rebuild the extension, stop on a build error, then run tools/gpu_quick.sh on an MI355X box
cd "$(dirname "$0")/.." || exit 1
python -c "
import sys; sys.path.insert(0,'.')
from orbslamm_amd import _lib; _lib.build(force=True)" > /tmp/build.log 2>&1 || { grep -E "error" -A7 /tmp/build.log | head -40; echo BUILD FAILED; exit 1; }
timeout 1500 /usr/local/graft/bin/gpurun --timeout 600 -- 'bash tools/gpu_quick.sh' 2>&1 | tail -5
