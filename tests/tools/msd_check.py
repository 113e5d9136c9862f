import os, sys, ctypes, time
sys.path.insert(0,'/root/repo')
import numpy as np
from pysubstringsearch_amd import _ffi
from oracle import oracle as O
import torch
lib=_ffi.lib
def build(host):
    n=host.size
    dT=torch.from_numpy(host).cuda(); dSA=torch.empty(n,dtype=torch.int32,device='cuda')
    st=_ffi.SaStats()
    _ffi.check(lib.pss_sa_build_device(dT.data_ptr(), dSA.data_ptr(), n, 0, 0, ctypes.byref(st)))
    return dSA.cpu().numpy(), st.as_dict()
os.environ['PSS_MSD']='1'
rng=np.random.default_rng(0)
bad=0
for trial in range(90):
    for k in ('PSS_MSD_NO_FUSE','PSS_MODE','PSS_MSD_SLOW_LOCAL'): os.environ.pop(k,None)
    if trial % 3 == 1: os.environ['PSS_MSD_NO_FUSE']='1'
    if trial % 5 == 2: os.environ['PSS_MODE']=['dense','sparse','text'][trial % 3]
    if trial % 7 == 3: os.environ['PSS_MSD_SLOW_LOCAL']='1' 
    n=int(rng.choice([2,3,17,100,4097,8192,8193,20000,70001,300000,1<<20,(1<<21)+77]))
    alpha=int(rng.choice([1,2,3,4,16,39,100,255]))
    t=rng.integers(0,alpha,n).astype(np.uint8)+ (0 if alpha>200 else 40)
    if rng.random()<0.3: t[rng.integers(0,n,max(1,n//50))]=10
    sa,st=build(t)
    ref=O.sa(t)
    ok=np.array_equal(sa,ref)
    print(trial,n,alpha,'msd',st['msd'],'maxb',st['msd_max_bucket'],'buckets',st['msd_buckets'],'tiles',st['msd_tiles'],'rounds',st['rounds'],'OK' if ok else 'FAIL',flush=True)
    bad+= (not ok)
for k in ('PSS_MSD_NO_FUSE','PSS_MODE','PSS_MSD_SLOW_LOCAL'): os.environ.pop(k,None)
# duplicates: big tied groups inside buckets (general local kernel + long groups in the emit)
line=bytes(rng.integers(97,123,200).astype(np.uint8))+b'\n'
for mode in (None,'dense'):
    if mode: os.environ['PSS_MODE']=mode
    t=np.frombuffer(line*3000,dtype=np.uint8).copy(); sa,st=build(t); ok=np.array_equal(sa,O.sa(t)); bad+=(not ok)
    print('dups',mode,'msd',st['msd'],'slow',st['msd_slow_tiles'],'rounds',st['rounds'],'OK' if ok else 'FAIL')
os.environ.pop('PSS_MODE',None)
for kind in (0,1,2,3):
    n=1<<22
    t=np.empty(n,np.uint8); lib.pss_gen_corpus(kind,t.ctypes.data,n,0)
    sa,st=build(t); ok=np.array_equal(sa,O.sa(t)); bad+=(not ok)
    print('corpus',kind,'msd',st['msd'],'maxb',st['msd_max_bucket'],'OK' if ok else 'FAIL')
print('BAD',bad)
os.environ.pop('PSS_MSD')
n=1<<29
t=np.empty(n,np.uint8); lib.pss_gen_corpus(0,t.ctypes.data,n,0)
dT=torch.from_numpy(t).cuda(); dSA=torch.empty(n,dtype=torch.int32,device='cuda'); st=_ffi.SaStats()
import bench
g=bench.load_big_goldens()[('lines',0,n)]
for flags in (0,0,0,1):
    for env in ('0', None):
        if env is None: os.environ.pop('PSS_MSD',None)
        else: os.environ['PSS_MSD']=env
        _ffi.check(lib.pss_sa_build_device(dT.data_ptr(), dSA.data_ptr(), n, 0, flags, ctypes.byref(st)))
        d=st.as_dict()
        print('PSS_MSD',env,'flags',flags,'ms',round(d['ms_total'],2),'msd',d['msd'],'maxb',d['msd_max_bucket'],'tiles',d['msd_tiles'],'g1',round(d['msd_ms_g1'],2),'g2',round(d['msd_ms_g2'],2),'loc',round(d['msd_ms_local'],2),'rounds',d['rounds'],'active',d['sum_active'],'verified',bench.sa_poly64_torch(dSA)==g['sa_poly64'],flush=True)
t=np.empty(n,np.uint8); lib.pss_gen_corpus(1,t.ctypes.data,n,0)
dT=torch.from_numpy(t).cuda()
_ffi.check(lib.pss_sa_build_device(dT.data_ptr(), dSA.data_ptr(), n, 0, 0, ctypes.byref(st)))
d=st.as_dict(); print('words ms',round(d['ms_total'],2),'msd',d['msd'],'maxb',d['msd_max_bucket'])
