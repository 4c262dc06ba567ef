#!/bin/bash
# round 6, session 6: general selection (8-bit digits, fused pick), incremental mirror / id table tests, failed tests of session 5
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${WAX_TAG:-r06_s6}
mkdir -p "$OUT"; cd "$R"; export TMPDIR=/tmp
timeout 400 python tools/general_select_bench.py --rows 1000000 --dims 384 --topk 10 195 300 1000 10000 --out "$OUT/general_select.jsonl" > /dev/null 2> "$OUT/gs.err"
timeout 600 python tools/general_select_bench.py --rows 10000000 --dims 384 --topk 10 195 300 1000 10000 --steps 30 --out "$OUT/general_select.jsonl" > /dev/null 2>> "$OUT/gs.err"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_gs" -o gs -- python $R/tools/general_select_bench.py --rows 10000000 --dims 384 --topk 300 --steps 20 > /dev/null 2>> "$OUT/gs.err")
find "$OUT/prof_gs" -name "*kernel_stats.csv" -exec cp {} "$OUT/gs_k300_10m_kernel_stats.csv" \;
rm -rf "$OUT/prof_gs"
cat "$OUT/general_select.jsonl" | cut -c1-230
cut -c1-180 "$OUT/gs_k300_10m_kernel_stats.csv" | head -8
timeout 2400 python -m pytest tests -m gpu -q -x -rf -p no:cacheprovider --timeout 500 -k "ingest or mirror or id_table or k_sweep or general or sharded_batched_device_resident or ticket_path_equals or bench_secondaries or filtered or concurrent_batched or serialize or upsert or remove" > "$OUT/pytest_sel.log" 2>&1; echo "pytest rc $?" >> "$OUT/pytest_sel.log"
tail -40 "$OUT/pytest_sel.log"
