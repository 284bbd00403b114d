"""The multi-rank code of tests/sharded_harness.py on the GPU box, with real collectives (-m gpu).  Every rank is a FRESH
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
    """`python bench.py --gpus 2 --rehearse` WITHOUT a launcher: bench.py starts its two ranks itself (a fresh
    torch.distributed.run child, before any HIP call) and relays rank 0's JSON line.  --rehearse: the box has one GPU and RCCL
    takes one rank per device, so rank 0 hosts both ranks of the kmg_group on cuda:0 (loopback exchange).  The N > 1 code path of
    the benchmark -- strong scaling of the BASELINE image through kmg_group_lloyd_*, the shapes tried before the timed region,
    weak-scaling and one-GPU extras -- must produce its line (the timings of a rehearsal mean nothing)."""
    import json
    root = os.path.dirname(HERE)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for name in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(name, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--rehearse"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert r.stdout.count("\n") == 1 and r.stdout.startswith("{"), r.stdout[:300]      # stdout carries the JSON line only
    line = json.loads(r.stdout)
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["metric"].endswith("(8192x8192, k=256)")
    cfg = line["config"]
    assert cfg["collective_ranks"] == 2 and "loopback" in cfg["collective_backend"] and "kmg_group_lloyd" in cfg["driver"]
    # the shapes of "one image over N GPUs" were tried and the fastest one measured
    choice = cfg["sharding_choice"]
    assert choice["cells_ms_per_step"] > 0 and choice["bands_ms_per_step"] > 0 and (choice["picked"] + "_ms_per_step") in choice
    assert ("cube pass sharded by cells" in cfg["sharding"]) == (choice["picked"] == "cells") and line["value"] > 0
    assert cfg["expected_speedup"] in (1.68, 1.35) and cfg["measured_speedup"] > 0 and cfg["one_gpu_ms_per_step_same_run"] > 0
    assert line["extra"]["weak_scaling_value"] > 0
    # a candidate that fails (as a broken collective would) costs the run that candidate only: the ranks drop the group, make a new
    # one and measure another shape
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--rehearse",
                        "--no-extras", "--no-cpu-baseline", "--fail-first-candidate"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    choice = json.loads(r.stdout)["config"]["sharding_choice"]
    assert choice["cells_ms_per_step"] is None and "fail-first-candidate" in choice["failed_on_this_rank"]["cells"]
    assert choice["picked"] != "cells" and choice[choice["picked"] + "_ms_per_step"] > 0
    # and the cell-sharded loop when asked for, started the way the driver starts N > 1
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--rehearse", "--cells", "--no-extras", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert "cube pass sharded by cells" in line["config"]["sharding"] and "sharding_choice" not in line["config"] and line["value"] > 0


def test_bench_one_rank_under_the_launcher_with_real_rccl(torch_cuda):
    """the driver's N > 1 start (torch.distributed.run, RANK / WORLD_SIZE from the environment) with ONE rank and --force-dist:
    the unique id made by rank 0, kmg_group_create_rank, and every per-iteration ncclAllReduce really issued by the library"""
    import json
    root = os.path.dirname(HERE)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                        "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
                        "--force-dist", "--no-extras", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["config"]["collective_backend"].startswith("RCCL 2") and line["config"]["collective_ranks"] == 1 and line["value"] > 0
    assert line["config"]["update"] == "k_update launch"
