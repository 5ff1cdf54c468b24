"""World-size-2 (and 3) gloo tests of the N>1 path on CPU: shards cover the batch once,
rank 0 reassembles the logits in global image order, timing is the max over ranks."""
import json
import os
import socket
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world,total", [(2, 8), (2, 7), (3, 10)])
def test_sharded_gather_over_gloo(tmp_path, world, total):
    port = _free_port()
    out = tmp_path / "rank0.json"
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(ROOT / "tests" / "dist_worker.py"), str(total), "1000",
                                       str(out)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    logs = [p.communicate(timeout=180)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)
    res = json.loads(out.read_text())
    assert res["ok"] and res["rows"] == total and res["world"] == world
    assert res["slowest"] == float(world)            # max over ranks of (1 + rank)
    assert sum(res["counts"]) == total
