import sys, os
ROOT="/root/repo"
for p in ("oracle","tests","iccv2025-upp_amd"): sys.path.insert(0, os.path.join(ROOT,p))
import numpy as np, torch
import _seeded, oracle
from models import build_model_from_cfg, upp_layers as L
from utils.config import builtin_cfg
from upp_hip import functional as HF
KEYS=['downstream_adapter', 'downstream_adapter1', 'downstream_prompts', 'dense_pred', 'mask_token', 'rectify_prompter','shape_pred', 'coarse_pred', 'predict_token_generator', 'mask_prompter', 'mask_token_generator']
pts, labels = _seeded.noisy_clouds(2, 1024, 0), torch.tensor([3, 17])
m = _seeded.fill(build_model_from_cfg(builtin_cfg('unify_modelnet_cls').model)).eval().cuda()
for n,p in m.named_parameters(): p.requires_grad_(any(k in n for k in KEYS))
lo=m(pts.cuda(), completion_prompt=True, denoise=True, point_num=1024); loss,_=m.get_loss_acc(lo, labels.cuda()); loss.backward()
prod={n:p.grad.detach().double().cpu() for n,p in m.named_parameters() if p.requires_grad and p.grad is not None}
m.zero_grad(set_to_none=True)
tr={'mode':'record','items':[]}; L.POOL_TRACE=tr
lo=m(pts.cuda(), completion_prompt=True, denoise=True, point_num=1024); loss,_=m.get_loss_acc(lo, labels.cuda()); loss.backward(); L.POOL_TRACE=None
hip={n:p.grad.detach().double().cpu() for n,p in m.named_parameters() if p.requires_grad and p.grad is not None}
print("recorded", [(k[0], k[1]) for k,_ in tr['items']])
ops_=oracle.torch_ops(); L.OPS.update(ops_); HF.fps_gather=ops_["fps_gather"]
def run64(replay, dt=torch.float64):
    m64 = _seeded.fill(build_model_from_cfg(builtin_cfg('unify_modelnet_cls').model)).eval()
    for n,p in m64.named_parameters(): p.requires_grad_(any(k in n for k in KEYS))
    m64=m64.to(dt)
    L.POOL_TRACE={'mode':'replay','items':tr['items']} if replay else None
    lo=m64(pts.to(dt), completion_prompt=True, denoise=True, point_num=1024); l,_=m64.get_loss_acc(lo, labels); l.backward()
    if replay: print("consumed", L.POOL_TRACE.get('pos'))
    L.POOL_TRACE=None
    return {n:p.grad.double() for n,p in m64.named_parameters() if p.requires_grad and p.grad is not None}
r1=run64(True); r0=run64(False); c32=run64(True, torch.float32)
def worst(a,b):
    out=[]
    for n in a:
        sc=b[n].abs().max().item(); out.append(((a[n]-b[n]).abs().max().item()/max(sc,1e-30), n))
    return sorted(out, reverse=True)[:6]
print("hip vs f64 gated  :", worst(hip,r1))
print("hip vs f64 ungated:", worst(hip,r0))
print("gated vs ungated  :", worst(r1,r0))
print("cpu f32 (torch) vs f64 gated:", worst(c32,r1))
print("hip vs cpu f32:", worst(hip,c32))
print("product (fused) vs cpu f32:", worst(prod,c32))
print("product (fused) vs hip traced:", worst(prod,hip))
