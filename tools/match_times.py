"""A/B of the two all-pairs kernels behind orbfe_match_bf_frames_device (GPU box): 256 pairs of 1004 x 1004 descriptors.
Prints one JSON object (kept as profiles/r02_match_variants.json)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from orb_slam2_ssd_semantic_amd import ORBmatcher, _ffi

B, cap, n = 256, 1088, 1004
rng = np.random.default_rng(0)
desc = torch.from_numpy(rng.integers(0, 256, (B, cap, 32), dtype=np.uint8)).cuda()
kps = torch.zeros((B, cap, 7), dtype=torch.float32, device="cuda")
dn = torch.full((B,), n, dtype=torch.int32, device="cuda")
qf = torch.arange(B, dtype=torch.int32, device="cuda")
tf = (qf + B - 1) % B
L = _ffi.lib()
out = {"pairs": B, "queries": n, "train": n, "distance_evaluations_per_call": B * n * n}
res = []
for kern, name in ((0, "k_match_bf (int8 MFMA, exact dot = 128*(128-d))"), (1, "k_match_popc (xor + v_bcnt_u32_b32)")):
    mt = ORBmatcher(0.9, True)
    mt.set_bf_kernel(kern)
    dm = torch.zeros((B, cap), dtype=torch.int32, device="cuda")
    nm = torch.zeros(B, dtype=torch.int32, device="cuda")

    def run():
        rc = L.orbfe_match_bf_frames_device(mt.handle, kps.data_ptr(), desc.data_ptr(), dn.data_ptr(), cap, qf.data_ptr(),
                                            tf.data_ptr(), B, 0.9, 100, 1, dm.data_ptr(), nm.data_ptr(), None)
        assert rc == 0
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 200
    out[name] = {"ms_per_call_incl_rot_prune": round(ms, 4), "Gdist_per_s": round(B * n * n / (ms * 1e-3) / 1e9, 1)}
    res.append(dm.cpu().numpy().copy())
out["identical_results"] = bool(np.array_equal(res[0], res[1]))
print(json.dumps(out))
