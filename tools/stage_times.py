import sys, json, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from orb_slam2_ssd_semantic_amd import ORBextractor
from bench import make_frames
B=int(os.environ.get("B","256")); w,h=640,480
ext=ORBextractor(1000,1.2,8,20,7,max_width=w,max_height=h,max_batch=B)
cap=ext.capacity()
fr=torch.from_numpy(make_frames(B,w,h,10000)).cuda()
dk=torch.zeros((B,cap,7),dtype=torch.int32,device="cuda"); dd=torch.zeros((B,cap,32),dtype=torch.uint8,device="cuda"); dn=torch.zeros(B,dtype=torch.int32,device="cuda")
for i in range(3): ext.extract_batch_device(fr.data_ptr(),B,w,h,w,w*h,dk.data_ptr(),dd.data_ptr(),cap,dn.data_ptr(),None)
torch.cuda.synchronize(); ext.set_profiling(True)
for i in range(10): ext.extract_batch_device(fr.data_ptr(),B,w,h,w,w*h,dk.data_ptr(),dd.data_ptr(),cap,dn.data_ptr(),None)
torch.cuda.synchronize()
print(os.environ.get("ORBFE_DEBUG","0"), {k:round(v,4) for k,v in ext.stage_ms().items()})
ext.set_profiling(False)
e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
for i in range(3): ext.extract_batch_device(fr.data_ptr(),B,w,h,w,w*h,dk.data_ptr(),dd.data_ptr(),cap,dn.data_ptr(),None)
torch.cuda.synchronize(); e0.record()
for i in range(20): ext.extract_batch_device(fr.data_ptr(),B,w,h,w,w*h,dk.data_ptr(),dd.data_ptr(),cap,dn.data_ptr(),None)
e1.record(); torch.cuda.synchronize()
print("split", os.environ.get("ORBFE_SPLIT","1"), "total_ms_unprofiled", round(e0.elapsed_time(e1)/20,4))
