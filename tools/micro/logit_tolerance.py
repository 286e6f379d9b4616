import sys, os, numpy as np
sys.path[:0]=['/root/repo','/root/repo/iccv2025-upp_amd','/root/repo/tests']
import torch, _seeded
from models import build_model_from_cfg
from utils.config import builtin_cfg
g=np.load('/root/repo/tests/golden/upp_model.npz')
m=_seeded.fill(build_model_from_cfg(builtin_cfg('unify_modelnet_cls').model)).eval().cuda()
with torch.no_grad():
    lc=m(_seeded.unit_ball_clouds(2,1024,0).cuda()).cpu().numpy()
    ln=m(_seeded.noisy_clouds(2,1024,0).cuda(), completion_prompt=True, denoise=True, point_num=1024).cpu().numpy()
for a,b,n in ((lc,g['logits_clean'],'clean'),(ln,g['logits_noisy'],'noisy')):
    print(n,'max abs',np.abs(a-b).max(),'scale',np.abs(b).max(),'max rel', (np.abs(a-b)/np.maximum(np.abs(b),1e-3)).max())
