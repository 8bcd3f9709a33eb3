"""N > 1 path on CPU: world_size 2 and 3 gloo jobs exercising the torch.distributed
transport used by the multi-GPU bench and the strip-partitioned relaxation algorithm."""
import os
import socket
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


from conftest import free_port  # noqa: E402


@pytest.mark.parametrize("world,periodic", [(2, 0), (2, 1), (3, 1)])
def test_gloo_strips(tmp_path, world, periodic):
    out = tmp_path / "ok.txt"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()), WORLD_SIZE=str(world),
               OMP_NUM_THREADS="1", GLOO_SOCKET_IFNAME="lo")     # (rendezvous on 127.0.0.1: gloo must not try to resolve the hostname)
    procs = []
    for r in range(world):
        e = dict(env, RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "_gloo_strip_worker.py"), str(periodic), str(out)],
                                      env=e, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    logs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            pytest.fail("gloo job timed out")
        logs.append(o.decode())
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)
    assert out.read_text() == "ok"
