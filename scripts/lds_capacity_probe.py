import torch, ctypes as C
hip=C.CDLL('libamdhip64.so')
v=C.c_int(0)
for name,idx in (('MaxSharedMemoryPerBlock',None),):
    pass
p=torch.cuda.get_device_properties(0)
print(p.name, 'shared_memory_per_block', getattr(p,'shared_memory_per_block',None), 'per_block_optin', getattr(p,'shared_memory_per_block_optin',None), 'multi_processor_count', p.multi_processor_count)
import sys; sys.path.insert(0,'pose-graph-initialization_amd')
from pyposegraphbuilder import Engine, synthetic as S
import numpy as np
eng=Engine()
for N in (2048, 2816, 3072, 4096, 8000, 8192, 9000):
    b=S.make_batch(np.arange(512),N)
    db=eng.upload(b['x1'],b['y1'],b['x2'],b['y2'],b['offsets'],7.5e-4,seed=1)
    eng.estimate_pose_batch(db); torch.cuda.synchronize()
    a,z=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    a.record(); eng.estimate_pose_batch(db); z.record(); torch.cuda.synchronize()
    print(N, '%.3f ms'%a.elapsed_time(z), '%.0f edges/s'%(512/a.elapsed_time(z)*1e3))
