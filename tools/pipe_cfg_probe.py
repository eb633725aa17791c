"""config 4 / config 5 shapes through the sequence pipeline (extract only) against one batched call on one handle"""
import json, os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from orb_slam2_ssd_semantic_amd import FramePipeline, ORBextractor  # noqa: E402

out = {}
st = torch.cuda.current_stream().cuda_stream
for name, (w, h, nf, N, seeds) in {"config5": (1920, 1080, 4000, 512, 64), "config4": (640, 480, 2000, 1024, 256)}.items():
    base = torch.from_numpy(bench.base_frames("S", seeds, w, h, 20000)).cuda()
    fr = bench.expand_frames(base, N)
    row = {}
    e = ORBextractor(nf, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=N)
    cap = e.capacity()
    k = torch.zeros((N, cap, 7), dtype=torch.int32, device="cuda"); d = torch.zeros((N, cap, 32), dtype=torch.uint8, device="cuda")
    n = torch.zeros(N, dtype=torch.int32, device="cuda")
    def one():
        e.extract_batch_device(fr.data_ptr(), N, w, h, w, w * h, k.data_ptr(), d.data_ptr(), cap, n.data_ptr(), st)
    for _ in range(3): one()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(10): one()
    torch.cuda.synchronize(); row["one_call_fps"] = round(10 * N / (time.perf_counter() - t), 1)
    ref = (k.clone(), d.clone(), n.clone())
    e.close(); del e; torch.cuda.empty_cache()
    for sub, pipes in ((N // 8, 8), (N // 16, 8), (N // 4, 4), (N // 16, 16)):
        pl = FramePipeline(nf, 1.2, 8, 20, 7, max_width=w, max_height=h, sub_batch=sub, npipes=pipes)
        def onep(flags=pl.NO_JOIN):
            pl.extract_match_device(fr.data_ptr(), N, w, h, w, w * h, k.data_ptr(), d.data_ptr(), cap, n.data_ptr(), None, None, flags=flags, stream=st)
        for _ in range(3): onep()
        pl.synchronize(); torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(10): onep()
        pl.synchronize(); torch.cuda.synchronize()
        row[f"pipeline_sub{sub}_pipes{pipes}_fps"] = round(10 * N / (time.perf_counter() - t), 1)
        row[f"pipeline_sub{sub}_pipes{pipes}_equal"] = bool(torch.equal(k, ref[0]) and torch.equal(d, ref[1]) and torch.equal(n, ref[2]))
        pl.close(); del pl; torch.cuda.empty_cache()
    out[name] = row
print(json.dumps(out))
