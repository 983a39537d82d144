import numpy as np, torch
from oracle import philox
from scasml_gp_amd import _lib
lib=_lib.load()
B,d=4096,100
out=torch.empty((B,d),dtype=torch.float32,device='cuda')
_lib.check(lib.scasml_debug_normals(_lib.Rng(0,0,0,0,1),0,d,B,_lib.ptr(out),_lib.stream_ptr()),'x')
got=out.cpu().numpy(); want=philox.normals(0,0,np.arange(B),0,d)
ne=(got.view(np.uint32)!=want.view(np.uint32))
print('mismatch',ne.sum(),'of',ne.size,'max abs',np.abs(got-want).max())
i=np.argwhere(ne)[:5]; print(i.tolist(), got[ne][:5], want[ne][:5])
print('by component mod 4:', [int(ne[:,c::4].sum()) for c in range(4)])
