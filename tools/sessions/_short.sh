set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=gpurun_out/r06_short5; mkdir -p $O; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider --timeout 400 > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
for n in 10000000 1000000 100000; do
  for t in 1 0 1 0; do
    timeout 600 python tools/general_select_bench.py --rows $n --dims 384 --topk 10 64 65 100 192 193 1000 4096 --steps 40 --tune select_short=$t --out $O/by_k.jsonl > /dev/null 2>> $O/err.txt
  done
done
timeout 200 python tools/fuzz_batch.py --seconds 100 --sharded 0.3 --seed 33 > $O/fuzz.txt 2>&1; tail -1 $O/fuzz.txt
