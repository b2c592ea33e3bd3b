import sys, os, ctypes as C, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from kami_amd import NN, weights as W, _lib as L
F,Cc,R,B=119,64,int(os.environ.get("RR","6")),512
nn=NN(8,8,F,4672,filters=Cc,residuals=R,dtype="bf16")
nn.load_weights(W.random_weights(F,Cc,R,seed=1,peaky=20.0),1)
x=np.random.default_rng(0).random((B,8,8,F),dtype=np.float32)
try:
    p,v=nn.infer(x)
except Exception as e:
    p=np.ones((1,1),np.float32); v=np.zeros(1,np.float32)
lib=L.load()
d_in=C.c_void_p(); d_p=C.c_void_p(); d_v=C.c_void_p()
lib.kh_dev_alloc(nn.handle, x.nbytes, C.byref(d_in)); lib.kh_dev_alloc(nn.handle, B*4672*4, C.byref(d_p)); lib.kh_dev_alloc(nn.handle, B*256*4, C.byref(d_v))
lib.kh_memcpy_h2d(nn.handle, d_in, x.ctypes.data_as(C.c_void_p), x.nbytes)
ms=C.c_float()
lib.kh_time_infer_device(nn.handle, d_in, B, d_p, d_v, 200, C.byref(ms))
print("DBG", os.environ.get("KAMI_TOWER_DBG"), "R", R, "ms", round(ms.value,5), "checksum", float(np.abs(np.log(p)).sum()), float(v.sum()))
