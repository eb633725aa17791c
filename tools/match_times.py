import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from orb_slam2_ssd_semantic_amd import ORBmatcher, _ffi
B=256; cap=1088; n=1004
rng=np.random.default_rng(0)
desc=torch.from_numpy(rng.integers(0,256,(B,cap,32),dtype=np.uint8)).cuda()
kps=torch.zeros((B,cap,7),dtype=torch.float32,device="cuda")
dn=torch.full((B,),n,dtype=torch.int32,device="cuda")
qf=torch.arange(B,dtype=torch.int32,device="cuda"); tf=(qf+B-1)%B
dm=torch.zeros((B,cap),dtype=torch.int32,device="cuda"); nm=torch.zeros(B,dtype=torch.int32,device="cuda")
mt=ORBmatcher(0.9,True); L=_ffi.lib()
def run():
    rc=L.orbfe_match_bf_frames_device(mt.handle,kps.data_ptr(),desc.data_ptr(),dn.data_ptr(),cap,qf.data_ptr(),tf.data_ptr(),B,0.9,100,1,dm.data_ptr(),nm.data_ptr(),None)
    assert rc==0
for i in range(3): run()
torch.cuda.synchronize()
e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(400): run()
e1.record(); torch.cuda.synchronize()
print("match_ms", round(e0.elapsed_time(e1)/400,4))
