"""bench.py must print its line on first contact with a multi-GPU node even when RCCL does not work: the
one-rank-per-GPU shape degrades to the gloo host exchange, the one-process shape to per-shard tickets, both labelled. CPU tests:
the decisions are host logic; the engines are faked."""
import json
import os
import socket
import subprocess
import sys
import types

import pytest

import bench

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("fake", [False, True])
def test_one_rank_per_gpu_shape_falls_back_to_gloo_when_rccl_is_unusable(fake, tmp_path):
    """Two ranks, no GPU: the mixed gloo + RCCL group either cannot be created or its probe fails (or the failure is injected); both
    ranks must agree on the host exchange, carry the reason, and the collectives bench.py runs next must work."""
    out = str(tmp_path / "init.json")
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WAX_OUT=out, OMP_NUM_THREADS="1")
        if fake:
            env["WAX_BENCH_FAKE_RCCL_FAILURE"] = "1"
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "_init_worker.py")], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    logs = [p.communicate(timeout=300)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)
    got = [json.load(open(out + f".{r}")) for r in range(2)]
    assert [g["use_rccl"] for g in got] == [False, False]
    assert all(g["why"] for g in got) and all(g["max"] == 2.0 for g in got)


class _FakeHandle:
    def __init__(self, ranks=None, raise_on_exchange=False):
        self.ranks, self.raise_on_exchange, self.set = ranks, raise_on_exchange, []

    def setTuning(self, key, value):  # noqa: N802
        if key == "exchange" and value == 1 and self.raise_on_exchange:
            raise RuntimeError("librccl.so: cannot open shared object file")
        self.set.append((key, value))

    def getTuning(self, key):  # noqa: N802
        return {"rccl_ranks": self.ranks}.get(key, 0)


@pytest.mark.parametrize("handle", [_FakeHandle(ranks=1), _FakeHandle(ranks=0), _FakeHandle(raise_on_exchange=True)])
def test_one_process_shape_degrades_to_tickets_and_still_prints_a_line(handle, monkeypatch):
    monkeypatch.delenv("WAX_BENCH_SAME_DEVICE", raising=False)
    monkeypatch.delenv("WAX_BENCH_FAKE_RCCL_FAILURE", raising=False)
    bench.RCCL_FAILURE[0] = None
    args = types.SimpleNamespace(exchange="rccl")
    mode, ranks = bench.handle_exchange(handle, args, 2)          # round 5: SystemExit — no line at all
    assert (mode, ranks) == (0, 0) and bench.RCCL_FAILURE[0]
    assert ("exchange", 0) in handle.set and ("ticket_path", 1) in handle.set
    full = {"metric": "m", "value": 1.0, "unit": "queries/s", "n_gpus": 2, "steps": 20, "warmup": 5, "ms_per_step": 1.0, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "w", "rows": 10, "dims": 384, "top_k": 10, "rccl_ranks": ranks, "shards": 2,
                       "exchange": f"in-library tickets (rccl failed: {bench.RCCL_FAILURE[0]})",
                       "preflight": {"ok": True, "peer_pairs": 2, "peer_enabled": 0, "devices": [0, 1]}},
            "secondary": [{"name": "h_rccl", "value": 1.0, "ms_per_step": 1.0, "n_gpus": 2, "rows_per_gpu": [5, 5],
                           "exchange": f"tickets (rccl failed: {bench.RCCL_FAILURE[0]})", "roofline": {"frac": 0.5, "kernel_avg_ms": 0.2}}]}
    line = json.loads(json.dumps(bench.compact_line(full), separators=(",", ":")))
    assert "rccl failed" in line["config"]["exchange"] and line["config"]["rccl_ranks"] == 0
    assert line["config"]["preflight"] == {"ok": True, "peer_pairs": 2, "peer_enabled": 0}
    assert "rccl failed" in line["secondary"][0]["exchange"]
    assert len(json.dumps(line, separators=(",", ":"))) < bench.LINE_BUDGET
    bench.RCCL_FAILURE[0] = None


def test_one_process_shape_keeps_rccl_when_it_spans_the_devices(monkeypatch):
    monkeypatch.delenv("WAX_BENCH_SAME_DEVICE", raising=False)
    monkeypatch.delenv("WAX_BENCH_FAKE_RCCL_FAILURE", raising=False)
    bench.RCCL_FAILURE[0] = None
    h = _FakeHandle(ranks=8)
    assert bench.handle_exchange(h, types.SimpleNamespace(exchange="rccl"), 8) == (1, 8) and bench.RCCL_FAILURE[0] is None
    assert bench.handle_exchange(_FakeHandle(ranks=8), types.SimpleNamespace(exchange="tickets"), 8) == (0, 0)
