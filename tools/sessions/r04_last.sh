#!/bin/bash
# round 4, last session: the full GPU suite and the end-of-round measurement set on the final build
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
export WAX_TAG=${WAX_TAG:-r04_final3}
OUT=$R/gpurun_out/$WAX_TAG
mkdir -p "$OUT"; cd "$R"; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -rf --durations=6 -p no:cacheprovider --timeout 400 > "$OUT/pytest_gpu.log" 2>&1
echo "pytest rc $?" >> "$OUT/pytest_gpu.log"
timeout 200 python __graft_entry__.py --smoke > "$OUT/smoke.log" 2>&1; echo "smoke rc $?" >> "$OUT/smoke.log"
bash tools/sessions/r04_final.sh
