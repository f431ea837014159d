import torch, sys
sys.path.insert(0, ".")
from act_amd.pointnet2_ops import pointnet2_utils as pu
import act_amd._C as C
x = torch.randn(128,1024,3,device="cuda")
for _ in range(3): i,c = pu.furthest_point_sample_with_centers(x,64)
C.prof_reset(); C.prof_enable(True)
for _ in range(20): i,c = pu.furthest_point_sample_with_centers(x,64)
torch.cuda.synchronize(); C.prof_enable(False)
t = C.prof_table()
print({k: round(v["ms"]/v["launches"]*1e3,1) for k,v in t.items()})
