"""bench.py's own multi-rank launch path, driven on CPU: `python bench.py --gpus 2` must re-execute itself under
torch.distributed.run with 2 ranks (gloo + the CPU stand-in extractor of --fake) and report n_gpus == 2 -- or refuse;
it must never print a 1-GPU line for an N-GPU request."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")
ARGS = ["--fake", "--steps", "2", "--warmup", "1", "--frames", "8", "--launches", "2", "--width", "64", "--height", "48"]


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    return env


def test_bench_spawns_its_own_ranks():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2"] + ARGS, env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout          # rank 0 prints ONE line
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["fake"] is True and j["steps"] == 2 and j["warmup"] == 1
    assert j["config"]["frames_per_gpu_per_step"] == 16
    assert j["gathered_frames"] == 2 * 16     # the per-step all-gather covered both ranks' batches
    ex = j["exchange"]                        # the exchange step's figures travel in the line (times are the GPU run's)
    assert ex["gather_bytes_per_rank"] == 16 * 4 + 16 * 64 * (28 + 32) and ex["ms_per_step_no_gather"] > 0 and ex["value_no_gather"] > 0
    assert j["scaling"] == "weak" and j["higher_is_better"] is True


def test_bench_refuses_a_wrong_world_size():
    env = _env()
    env.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2"] + ARGS, env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 2 and "does not match --gpus" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_bench_under_the_drivers_launcher():
    """the exact form the driver uses for N > 1"""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), BENCH, "--gpus", "2"] + ARGS
    r = subprocess.run(cmd, env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert j["n_gpus"] == 2
