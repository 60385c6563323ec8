"""CPU, world_size 2 and 8 over gloo: the N>1 sharding + score all-gather path of bench.py / test.py (trainer_ddp.py:144,259-267;
world 8 = C4's 900 videos over the 8 GPUs of one node, which no hardware run of this build has exercised)."""
import os
import socket
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import sys, torch
    sys.path.insert(0, %r)
    import kvq_amd
    from kvq_amd import dist as kd
    rank, local_rank, world = kd.init(backend="gloo")
    for n in (900, 7, 2, 1, 113):
        idx = kd.shard_indices(n, rank, world)
        assert len(idx) == -(-n // world)
        local = torch.tensor([float(i) * 0.5 + 1.0 for i in idx])      # score of item i = i/2 + 1
        full = kd.gather_scores(local, n, rank, world)
        assert torch.equal(full, torch.arange(n, dtype=torch.float32) * 0.5 + 1.0), (n, full)
    t = kd.max_over_ranks(1.0 + rank, "cpu")
    assert t == float(world)
    kd.barrier()
    print("OK", rank)
""") % ROOT


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world", [2, 8])
def test_shard_and_gather_ranks_gloo(tmp_path, world):
    """``videos[rank::world]`` with the wrap-around padding of DistributedSampler(shuffle=False) and ONE all_gather of the score vectors:
    900 (C4), sizes below the world size (7, 2, 1: ranks that hold only padding) and a non-multiple (113) come back in video order."""
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), WORLD_SIZE=str(world), OMP_NUM_THREADS="1")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = [p.communicate(timeout=480)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"OK {r}" in o, o[-2000:]


def test_single_process_is_identity():
    import torch
    import kvq_amd  # noqa: F401
    from kvq_amd import dist as kd
    assert kd.shard_indices(5, 0, 1) == [0, 1, 2, 3, 4]
    x = torch.arange(5, dtype=torch.float32)
    assert torch.equal(kd.gather_scores(x, 5, 0, 1), x)


@pytest.mark.parametrize("n", [2, 8])
def test_bench_gpus_n_without_rank_environment_launches_n_ranks_or_fails(n):
    """``python bench.py --gpus N`` with no RANK / WORLD_SIZE must never come back as a one-GPU run: it re-executes itself under
    torch.distributed.run with two ranks (which, on this GPU-less box, both fail on "needs a GPU": a non-zero exit and no
    result line), and a rank environment that contradicts --gpus is refused."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(n), "--steps", "1", "--warmup", "0"], env=env,
                       capture_output=True, text=True, timeout=600)
    lines = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert all(ln.get("n_gpus") == n for ln in lines), r.stdout
    assert lines or r.returncode != 0, (r.returncode, r.stdout[-500:], r.stderr[-500:])
    if not lines:      # no GPU here: the ranks were started (torch.distributed.run reports its failed children) and said why
        assert "ChildFailedError" in r.stderr or "needs a GPU" in r.stderr, r.stderr[-800:]
    if n != 2:
        return
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4", "--steps", "1", "--warmup", "0"],
                       env=dict(env, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1"), capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "WORLD_SIZE=1 but --gpus 4" in r.stderr and not r.stdout.strip()
