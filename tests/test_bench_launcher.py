"""``python bench.py --gpus N`` must really run N ranks (reference: one process per GPU + gradient averaging,
main_linprobe.py:581-583, util/misc.py:214-257).  CPU only: the ``--rendezvous-only`` mode runs the launcher, the
127.0.0.1 rendezvous and one all-reduce of a gradient-sized flat buffer over gloo -- no kernels."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(kw)
    return env


def _last_json(out: str) -> dict:
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out            # exactly ONE line, printed by rank 0
    return json.loads(lines[0])


def test_gpus_2_spawns_two_ranks():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--rendezvous-only"], env=_env(), capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = _last_json(r.stdout)
    assert line["n_gpus"] == 2 and line["rccl_ranks"] == 2 and line["allreduce_ok"] is True


def test_gpus_1_is_one_rank():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--rendezvous-only"], env=_env(), capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert _last_json(r.stdout)["n_gpus"] == 1


def test_world_size_mismatch_fails_loudly():
    """A launcher that started another number of ranks than --gpus says: exit non-zero instead of printing a line with
    the wrong n_gpus (round 1's bench ignored --gpus)."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "4", "--rendezvous-only"],
                       env=_env(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"), capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "WORLD_SIZE=1" in r.stderr


def test_under_torch_distributed_run():
    """The driver's launch form: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), BENCH, "--gpus", "2", "--rendezvous-only"],
                       env=_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = _last_json(r.stdout)
    assert line["n_gpus"] == 2 and line["rccl_ranks"] == 2


def test_global_batch_is_strong_scaling():
    """``--global-batch G``: the protocol's fixed effective batch (reference README.md:119-120: 4096) split over the ranks --
    per-GPU batch G / N, lr = 0.1 * G / 256 whatever N is (main_linprobe.py:572-573), and the line says "strong"."""
    for n, per_gpu in ((1, 4096), (2, 2048)):
        r = subprocess.run([sys.executable, BENCH, "--gpus", str(n), "--global-batch", "4096", "--rendezvous-only"], env=_env(),
                           capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        line = _last_json(r.stdout)
        assert line["scaling"] == "strong" and line["batch_per_gpu"] == per_gpu and line["global_batch"] == 4096
        assert abs(line["lr"] - 1.6) < 1e-12
    # the default stays weak scaling at 1024 per GPU
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--rendezvous-only"], env=_env(), capture_output=True, text=True, timeout=300)
    line = _last_json(r.stdout)
    assert line["scaling"] == "weak" and line["batch_per_gpu"] == 1024 and line["global_batch"] == 2048
    # a global batch the ranks cannot split evenly is refused
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--global-batch", "4097", "--rendezvous-only"], env=_env(),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
