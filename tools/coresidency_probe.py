"""Developer probe: does HBM-bound work with < 32 VGPRs run in the shadow of k_fast_map (3 waves x 160 registers per SIMD leave
exactly 32)?  Stream A: the extractor on 1024 frames, repeated; stream B: a filler kernel (tools/ubench/filler.hip, 10-12 VGPRs)
moving the pyramid's bytes (1.6 GB per 1024 frames), with 0 / 10 / 40 integer operations per 16 bytes, repeated.
Reports the times alone and together."""
import ctypes as C, json, os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from orb_slam2_ssd_semantic_amd import ORBextractor  # noqa: E402
from orb_slam2_ssd_semantic_amd.synth import synth_frames_parallel  # noqa: E402

F, w, h, REP = 1024, 640, 480, 12
L = C.CDLL(os.path.join(ROOT, "tools", "ubench", "libfiller.so"))
L.filler_copy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
L.filler_work.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p]
base = torch.from_numpy(synth_frames_parallel("S", 128, h, w, 10000)).cuda()
g = base.repeat(F // 128, 1, 1).contiguous()
e = ORBextractor(1000, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=F)
cap = e.capacity()
k = torch.zeros((F, cap, 7), dtype=torch.int32, device="cuda"); d = torch.zeros((F, cap, 32), dtype=torch.uint8, device="cuda")
n = torch.zeros(F, dtype=torch.int32, device="cuda")
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
nbytes = 800 * 1024 * 1024          # read 0.8 GB + write 0.8 GB = the pyramid pass's 1.6 GB per 1024 frames
src = torch.empty(nbytes, dtype=torch.uint8, device="cuda"); dst = torch.empty(nbytes, dtype=torch.uint8, device="cuda")

def run_A():
    for _ in range(REP):
        e.extract_batch_device(g.data_ptr(), F, w, h, w, w * h, k.data_ptr(), d.data_ptr(), cap, n.data_ptr(), sA.cuda_stream)
def run_B(ops, blocks):
    for _ in range(REP):
        if ops == 0: L.filler_copy(src.data_ptr(), dst.data_ptr(), nbytes, blocks, sB.cuda_stream)
        else: L.filler_work(src.data_ptr(), dst.data_ptr(), nbytes, blocks, ops, sB.cuda_stream)
def timed(fn):
    fn(); torch.cuda.synchronize(); t = time.perf_counter(); fn(); torch.cuda.synchronize(); return (time.perf_counter() - t) / REP * 1e3
out = {"extract_alone_ms": round(timed(run_A), 4)}
for ops in (0, 10, 40):
    for blocks in (1024, 4096):
        tb = timed(lambda: run_B(ops, blocks))
        both = timed(lambda: (run_A(), run_B(ops, blocks)))
        out[f"ops{ops}_blocks{blocks}"] = {"filler_alone_ms": round(tb, 4), "both_ms": round(both, 4), "serial_ms": round(out["extract_alone_ms"] + tb, 4)}
print(json.dumps(out))
