"""The multi-rank code of kmeans_gpu_amd.sharded on the GPU box, with real collectives (-m gpu).  Every rank is a FRESH
child process (started before it touches the GPU: tests/dist_child.py) -- one rank through RCCL, and two / three ranks that
share the box's one GPU through gloo -- and compares ShardedLloyd (row bands; cells=True) with the unsharded loop."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(backend, world):
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        env.pop("KMG_STRATEGY", None)
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "dist_child.py"), backend], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=420)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            pytest.fail(f"{backend} world {world}: a rank did not finish")
        outs.append(out)
    for rank, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"{backend} world {world} rank {rank} failed:\n{out[-3000:]}"
        assert "equal the unsharded loop" in out


def test_one_rank_rccl_group_runs_every_collective_of_the_sharded_loops(torch_cuda):
    _run("nccl", 1)


@pytest.mark.parametrize("world", [2, 3])
def test_ranks_sharing_one_gpu_over_gloo_equal_the_unsharded_loop(torch_cuda, world):
    _run("gloo", world)
