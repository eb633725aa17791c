"""Every N > 1 code path of bench.py with REAL kernels on a one-GPU box: two ranks share GPU 0
(`--same-device`), the per-step exchange goes through the host-staged gloo transport (RCCL refuses two ranks per device), the
library's pipeline joins the launch stream at the end of every step, the gather of step k overlaps the kernels of step k + 1
(two output sets), config 4 runs strong + weak on an UNEVEN global batch (1023), the guarded C-ABI group leg must come back --
with a result or with the error RCCL raises for two ranks on one device -- without hanging, and frames of the OTHER rank's
shard, read from the gathered blocks, equal the oracle.  The line is marked same_device and is never a scaling figure."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    env["ORBFE_BENCH_GROUP_TIMEOUT"] = "120"
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return env


@pytest.mark.gpu
def test_two_ranks_on_one_device_run_every_multi_gpu_path():
    cmd = [sys.executable, BENCH, "--gpus", "2", "--same-device", "--steps", "3", "--warmup", "1", "--frames", "128", "--launches", "4",
           "--seeds", "128", "--config4-batch", "1023"]
    r = subprocess.run(cmd, env=_env(), capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["same_device"] is True and "SAME-DEVICE TEST MODE" in j["config"]["workload"]
    assert j["exact_checked"] is True                       # rank 0's own frames of the timed region, incl. cross-sub-batch matches
    assert j["gathered_exact_checked"] is True              # rank 1's frames as they arrived through the overlapped gather
    assert {c["rank"] for c in j["gathered_exact_checked_frames"]} == {1}
    assert "joins the launch stream" in j["config"]["streams"]
    c4 = j["config4"]
    assert c4["global_batch"] == 1023 and c4["n_gpus"] == 2
    assert c4["strong"]["frames_per_gpu"] == 512 and c4["strong"]["gathered_frames"] == 2 * 512   # 512 + 511, padded to 2 x 512
    assert c4["weak"]["gathered_frames"] == 2 * 1023
    assert c4["gathered_exact_checked"] is True
    cg = c4["cabi_group"]
    # the library's own communicator: two ranks on one device is what RCCL refuses -- the leg reports that, the line survives
    assert ("frames_per_s" in cg) or ("error" in cg and cg["error"]), cg
    assert j["config4_cabi_group_status"]
    assert j["value"] > 0 and j["scaling"] == "weak"
