"""world_size-2 (and 3) multi-process tests of the sharded path on CPU with the gloo backend."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import oracle

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(world, n, d, k, pattern, tmp_path):
    out = str(tmp_path / f"out_{world}_{pattern}.json")
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   WAX_N=str(n), WAX_D=str(d), WAX_K=str(k), WAX_PATTERN=pattern, WAX_OUT=out, OMP_NUM_THREADS="2")
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "_gloo_worker.py")], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    logs = [p.communicate(timeout=600)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)
    return json.load(open(out))


@pytest.mark.parametrize("world,pattern", [(2, "gauss"), (2, "ties"), (3, "gauss")])
def test_sharded_exchange_matches_single_corpus_oracle(world, pattern, tmp_path):
    n, d, k = 5000, 64, 10
    got = _run(world, n, d, k, pattern, tmp_path)
    corpus = oracle.gaussian_unit_rows(0, n, d) if pattern == "gauss" else oracle.tie_pattern(0, n, d)
    queries = oracle.gaussian_unit_queries(3, d)
    if pattern != "gauss":
        queries = np.abs(queries)
    for qi, q in enumerate(queries):
        ids, scores, _, rows = oracle.search(0, corpus, np.arange(n, dtype=np.uint64) + 1000, q, k)
        assert got[qi]["ids"] == [int(x) for x in ids]  # identical ids at every shard count, ties included
        assert np.array_equal(np.asarray(got[qi]["scores"], dtype=np.float32), scores)


def test_more_ranks_than_rows(tmp_path):
    got = _run(2, 3, 8, 10, "gauss", tmp_path)  # rank 1's shard is empty, k > n
    corpus = oracle.gaussian_unit_rows(0, 3, 8)
    q = oracle.gaussian_unit_queries(3, 8)[0]
    ids, scores, _, _ = oracle.search(0, corpus, np.arange(3, dtype=np.uint64) + 1000, q, 10)
    assert got[0]["ids"] == [int(x) for x in ids] and len(ids) == 3
