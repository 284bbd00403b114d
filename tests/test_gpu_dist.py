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


def test_bench_multi_gpu_path_rehearsed_on_one_gpu(torch_cuda):
    """`bench.py --gpus 2` as the driver launches it (torch.distributed.run, one rank per process) with --rehearse: both ranks on
    cuda:0, collectives through gloo.  The N > 1 code path of the benchmark -- strong scaling of the BASELINE image, cell-sharded
    cube pass, weak-scaling extra -- must produce its JSON line (the timings of a rehearsal mean nothing)."""
    import json
    root = os.path.dirname(HERE)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("KMG_STRATEGY", None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--rehearse"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["metric"].endswith("(8192x8192, k=256)")
    # both shapes of "one image over N GPUs" were tried and the faster one measured
    choice = line["config"]["sharding_choice"]
    assert choice["cells_ms_per_step"] > 0 and choice["bands_ms_per_step"] > 0 and choice["picked"] in ("cells", "bands")
    assert ("cube pass sharded by cells" in line["config"]["sharding"]) == (choice["picked"] == "cells") and line["value"] > 0
    assert line["extra"]["weak_scaling_value"] > 0
    # and the cell-sharded loop when asked for
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--rehearse", "--cells", "--no-extras", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert "cube pass sharded by cells" in line["config"]["sharding"] and "sharding_choice" not in line["config"] and line["value"] > 0
