#!/bin/bash
# round 6, session 8: general selection with vectorised passes — timing, kernel rows, the tests that use it; bimodality probe on this box
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${WAX_TAG:-r06_s8}
mkdir -p "$OUT"; cd "$R"; export TMPDIR=/tmp
timeout 400 python tools/general_select_bench.py --rows 1000000 --dims 384 --topk 10 195 300 1000 10000 --out "$OUT/general_select.jsonl" > /dev/null 2> "$OUT/gs.err"
timeout 600 python tools/general_select_bench.py --rows 10000000 --dims 384 --topk 10 195 300 1000 10000 --steps 30 --out "$OUT/general_select.jsonl" > /dev/null 2>> "$OUT/gs.err"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_gs" -o gs -- python $R/tools/general_select_bench.py --rows 10000000 --dims 384 --topk 300 --steps 20 > /dev/null 2>> "$OUT/gs.err")
find "$OUT/prof_gs" -name "*kernel_stats.csv" -exec cp {} "$OUT/gs_k300_10m_kernel_stats.csv" \;
rm -rf "$OUT/prof_gs"
python -c "
import json
for l in open('$OUT/general_select.jsonl'):
    d=json.loads(l); print(d['rows'],d['top_k'],'pip %.4f blk %.4f %s'%(d['ms_pipelined'],d['ms_blocking'],d['checksum']))"
cut -c1-200 "$OUT/gs_k300_10m_kernel_stats.csv" | grep "select\|scan_kernel\|rank\|keys_to"
timeout 1800 python -m pytest tests -m gpu -q -x -rf -p no:cacheprovider --timeout 500 -k "k_sweep or general or sharded_batched_device_resident or ticket_path_equals or filtered or special or ties or dup or reference_cases or host_staged" > "$OUT/pytest_sel.log" 2>&1; echo "pytest rc $?" >> "$OUT/pytest_sel.log"
tail -5 "$OUT/pytest_sel.log"
for i in $(seq 0 7); do
WAX_PROBE_PAD_KB=$((i * 1216)) timeout 200 python tools/bimodal_probe.py --rows 1000000 --out "$OUT/bimodal.jsonl" > /dev/null 2>> "$OUT/bimodal.err"
done
python -c "
import json
for l in open('$OUT/bimodal.jsonl'):
    d=json.loads(l); print(d['store_ptr'], d['us_per_query_pipelined'][1:], d['us_scan_kernel_back_to_back'])"
