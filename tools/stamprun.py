import sys, os, ctypes as C, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
os.environ["KAMI_TOWER_DBG"] = "1024"
from kami_amd import NN, weights as W, _lib as L
F,Cc,R,B=119,64,int(os.environ.get("RR","6")),512
nn=NN(8,8,F,4672,filters=Cc,residuals=R,dtype="bf16")
nn.load_weights(W.random_weights(F,Cc,R,seed=1,peaky=20.0),1)
x=np.random.default_rng(0).random((B,8,8,F),dtype=np.float32)
for it in range(3):
    p,vf,lg=nn.infer_full(x)
st=lg.reshape(-1).view(np.uint64)[:4*256*3].reshape(4,256,3).astype(np.int64)
n=18+18*R+6
t0=st[0,0,0]
print("step  w0:arrive vmwait barwait | per-wave arrive deltas (cycles @100MHz*? raw memtime units)")
for i in range(n):
    row=[]
    for w in range(4):
        row.append("%6d %4d %4d"%(st[w,i,0]-t0, st[w,i,1]-st[w,i,0], st[w,i,2]-st[w,i,1]))
    print("%3d | "%i + " | ".join(row))
# boundary stamps: [wave][boundary][point]: 0 = all MFMAs of the layer issued, 1 = epilogue + image writes issued, 2 = own writes complete
bs=lg.reshape(-1).view(np.uint64)[4*256*3:4*256*3+4*64*4].reshape(4,64,4).astype(np.int64)
print("boundary (layer index in tower) | wave0: last-step arrive -> mfma issued -> epilogue issued -> lgkm done -> next step arrive (deltas)")
for b in range(2*R):
    last=18+9*b+8; nxt=last+1
    row=[]
    for w in (0,2):
        a0=st[w,last,0]; pts=[bs[w,b,0],bs[w,b,1]]+([bs[w,b,2]] if b%2==0 else [])+[st[w,nxt,0]]
        d=np.diff(np.array([a0]+pts))
        row.append(" ".join("%5d"%x for x in d))
    print("%2d | "%b + " | ".join(row))
d=np.diff(st[0,:n,0])
print("median step (wave0) =", np.median(d), " tower median =", np.median(d[18:18+18*R]))
print("total steps span", st[0,n-1,2]-st[0,0,0])
