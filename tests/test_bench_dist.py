"""CPU, 2 processes over gloo: the multi-replica bookkeeping of bench.py — ranks are independent
replicas (no data-path collective), the job time is the slowest rank's, the value is the whole job's
tokens over that time."""
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import bench
    assert bench.dist_env() == (rank, world, rank)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dist.barrier()
    elapsed = 1.0 + 0.5 * rank                      # rank 1 is the slow replica
    job = bench.max_over_ranks(elapsed, world)
    tokens = bench.whole_job_tokens(8, 100, world)
    out.put((rank, job, tokens / job))
    dist.barrier()
    dist.destroy_process_group()


def test_two_replicas_gloo():
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(out.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, job, value in res:
        assert job == 1.5                            # max over ranks on every rank
        assert abs(value - 16 * 8 * 100 * 2 / 1.5) < 1e-6


def test_single_process_defaults(monkeypatch):
    sys.path.insert(0, ROOT)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    import bench
    assert bench.dist_env() == (0, 1, 0)
    assert bench.max_over_ranks(2.5, 1) == 2.5
    assert bench.whole_job_tokens(8, 10, 1) == 1280
    cfg = __import__("nano_qwen3tts_vllm.utils.synthetic", fromlist=["x"]).make_config("1.7B")
    b = bench.algorithmic_bytes_per_frame(cfg, 8, 300)
    assert 5.4e9 < b < 5.8e9                         # 2.83 GB talker + 15 x 0.166 GB predictor + KV rows


def test_gpus_flag_starts_its_own_ranks_and_reports_a_failed_one():
    """`python bench.py --gpus 2` with no launcher around it starts two ranks itself (RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_* exported per child).  There is no GPU here, so every rank fails at its first GPU call: the launcher must
    stop the others and exit non-zero, naming the rank - it must not hang or report success."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    if torch.cuda.is_available():
        env["HIP_VISIBLE_DEVICES"] = ""            # on a GPU box: hide the devices, same outcome
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "rank" in r.stderr and "exited with" in r.stderr
    assert r.stdout.strip() == ""                  # no JSON line from a failed job
