import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from orb_slam2_ssd_semantic_amd import ORBextractor
from bench import make_frames
B=256; w,h=640,480
ext=ORBextractor(1000,1.2,8,20,7,max_width=w,max_height=h,max_batch=B)
cap=ext.capacity()
fr=torch.from_numpy(make_frames(B,w,h,10000)).cuda()
dk=torch.zeros((B,cap,7),dtype=torch.int32,device="cuda"); dd=torch.zeros((B,cap,32),dtype=torch.uint8,device="cuda"); dn=torch.zeros(B,dtype=torch.int32,device="cuda")
for i in range(3): ext.extract_batch_device(fr.data_ptr(),B,w,h,w,w*h,dk.data_ptr(),dd.data_ptr(),cap,dn.data_ptr(),None)
torch.cuda.synchronize()
s=ext.selected(7,frame=0)
u=(s[:,0].astype(np.int64)+s[:,1].astype(np.int64)*4096+s[:,2].astype(np.int64)*(1<<24))
n=int(u[0]); ts=u[1:1+n]
print("n_ts",n,"keys",u[60],"S",u[61])
print("cycles(100MHz ticks?)",ts.tolist())
print("deltas",np.diff(ts).tolist())
