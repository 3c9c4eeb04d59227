import sys, os, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from threadpoolctl import threadpool_limits, threadpool_info
from oracle import model as om
print(os.cpu_count(), [ (i['internal_api'], i['num_threads']) for i in threadpool_info()])
L,H,D,C,B,U = 3,512,40,80,32,161
for thr in (8, 16, 32, 64, 128, 256):
    with threadpool_limits(limits=thr):
        rng = np.random.RandomState(0)
        p = om.init_params(L,H,D,C,seed=1234,dtype=np.float32)
        m = {k: np.zeros_like(v) for k,v in p.items()}; v = {k: np.zeros_like(vv) for k,vv in p.items()}
        T=64
        x = rng.randn(T,B,D).astype(np.float32); lengths = np.full(B,T,np.int32)
        dense = np.zeros((B,U),np.int32); dense[:,0:5]=rng.randint(1,79,size=(B,5)); dense[:,5]=79
        t0=time.time(); om.train_step(p,m,v,1,[(x,lengths,dense)],L,3e-4,1.0); dt=time.time()-t0
        print('threads',thr,'frames/s',B*T/dt, 'time',dt)
