"""torch.distributed's RCCL backend on the one GPU of the box, world size 1: the exact process-group initialisation of
bench.py's N > 1 branch (backend "nccl" = RCCL, `device_id` bound, rendezvous on 127.0.0.1) and the exact collective of
`OverlappedKeyframeGather.launch` (asynchronous `all_gather_into_tensor` of the int32 count / int32-cast keypoint / uint8 descriptor
blocks the library's pipeline wrote, behind the pipeline's kernels on the current stream) run once for real: RCCL loads, a
communicator exists, the dtypes and shapes of the three blocks are accepted, the work handles order the streams; then bench.py's own
exchange object (`OverlappedKeyframeGather` on its RCCL transport: gather stream, event timing, valid-prefix blocks) through five
overlapped steps.  What a world of
one cannot show -- xGMI transport, several ranks -- stays with the world-2 gloo tests and the driver's multi-GPU run.  Runs in a
child process (a process group is process-wide state)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import os, socket, sys
sys.path.insert(0, %(root)r)
import numpy as np
import torch
import torch.distributed as dist
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from orb_slam2_ssd_semantic_amd.pipeline import FramePipeline
from orb_slam2_ssd_semantic_amd.synth import synth_tum_like
F, w, h = 6, 320, 240
pl = FramePipeline(500, 1.2, 8, 20, 7, max_width=w, max_height=h, sub_batch=4, npipes=2)
cap = pl.capacity()
g = torch.from_numpy(np.stack([synth_tum_like(70 + i, h, w) for i in range(F)])).cuda()
g_in = g
kps = torch.zeros((F, cap, 7), dtype=torch.int32, device="cuda")
desc = torch.zeros((F, cap, 32), dtype=torch.uint8, device="cuda")
n = torch.zeros(F, dtype=torch.int32, device="cuda")
match = torch.full((F, cap), -1, dtype=torch.int32, device="cuda")
nm = torch.zeros(F, dtype=torch.int32, device="cuda")
st = torch.cuda.current_stream().cuda_stream
pl.extract_match_device(g.data_ptr(), F, w, h, w, w * h, kps.data_ptr(), desc.data_ptr(), cap, n.data_ptr(), match.data_ptr(),
                        nm.data_ptr(), stream=st)     # joins the launch stream: the collective below is ordered behind it
outs, works = [], []
for t in (n, kps, desc):
    o = torch.empty_like(t)
    works.append(dist.all_gather_into_tensor(o, t.contiguous(), async_op=True))
    outs.append(o)
for wk in works:
    wk.wait()
torch.cuda.synchronize()
assert all(torch.equal(o, t) for o, t in zip(outs, (n, kps, desc)))
assert int(n.min()) > 50
# ... and bench.py's own exchange object on the RCCL transport: the collectives of a launch issued under the object's gather
# stream behind an event of the producer, HIP events around them (timing()), the valid prefix only (gather_cap), the producer
# re-using a set only after acquire()
from orb_slam2_ssd_semantic_amd.distributed import OverlappedKeyframeGather
sets = [(n.clone(), kps.clone(), desc.clone()) for _ in range(2)]
gcap = (int(n.max()) + 63) // 64 * 64
assert gcap < cap
g = OverlappedKeyframeGather(sets, gather_cap=gcap)
assert g.comm is not None and not g.host_staged and g.bytes_per_rank == F * 4 + F * gcap * (28 + 32)
for i in range(5):
    k = i & 1
    g.acquire(k)
    pl.extract_match_device(g_in.data_ptr(), F, w, h, w, w * h, sets[k][1].data_ptr(), sets[k][2].data_ptr(), cap,
                            sets[k][0].data_ptr(), match.data_ptr(), nm.data_ptr(), stream=st)
    g.launch(k)
for k in (0, 1):
    gn, gk, gd = g.result(k)
    torch.cuda.synchronize()
    assert torch.equal(gn, n) and torch.equal(gk, kps[:, :gcap]) and torch.equal(gd, desc[:, :gcap]) and g.truncated(k) == 0
tm = g.timing()
assert tm["launches"] == 5 and tm["mean_ms"] > 0
g.close()
dist.barrier()
dist.destroy_process_group()
pl.close()
print("RCCL_WORLD1_OK", int(n.sum()), round(tm["mean_ms"], 4))
'''


@pytest.mark.gpu
def test_rccl_process_group_and_the_gather_of_the_three_blocks_world_1():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}], cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       text=True, timeout=600)
    assert r.returncode == 0 and "RCCL_WORLD1_OK" in r.stdout, r.stdout[-3000:]
