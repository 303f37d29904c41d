import sys, time; sys.path.insert(0,'.')
import torch, numpy as np
from collab_splats_amd.rendering import rasterization
from collab_splats_amd.synthetic import random_scene
from oracle.craster import CRaster
def err(x,y):
    x=np.asarray(x,dtype=np.float64); y=np.asarray(y,dtype=np.float64); return float(np.abs(x-y).max()/max(1e-30,np.abs(y).max()))
for (N,W,H,mode,rm) in ((2000,256,256,"antialiased","RGB+ED"),(20000,640,360,"classic","RGB")):
    sc=random_scene(N,W,H,seed=42)
    dev='cuda'
    ins={k:v.to(dev) for k,v in sc.items()}
    means=ins['means'].requires_grad_(True); quats=ins['quats'].requires_grad_(True)
    ls=ins['log_scales'].requires_grad_(True); ol=ins['opacity_logits'].requires_grad_(True); sh=ins['sh'].requires_grad_(True)
    r,a,ed,md,n,meta=rasterization(means,quats,torch.exp(ls),torch.sigmoid(ol),sh,ins['viewmats'],ins['Ks'],W,H,sh_degree=3,render_mode=rm,rasterize_mode=mode,return_depth_normal=True,absgrad=True)
    torch.cuda.synchronize()
    print(N,W,H,mode,rm,'I',meta['n_isects'],'alpha mean',a.mean().item())
    cr=CRaster(np.float32)
    scales=torch.exp(sc['log_scales']).numpy(); op=torch.sigmoid(sc['opacity_logits']).numpy()
    st=cr.forward(sc['means'].numpy(),sc['quats'].numpy(),scales,op,sc['sh'].numpy(),sc['viewmats'][0].numpy(),sc['Ks'][0].numpy(),W,H,sh_degree=3,render_mode=rm,rasterize_mode=mode)
    print(' radii eq',np.array_equal(st['proj']['radii'],meta['radii'][0].cpu().numpy()),
      'depth bits eq',np.array_equal(st['proj']['depths'].view(np.uint32),meta['depths'][0].detach().cpu().numpy().view(np.uint32)),
      'means2d eq',np.array_equal(st['proj']['means2d'],meta['means2d'][0].detach().cpu().numpy()),
      'isect eq',np.array_equal(st['bins']['isect_ids'],meta['isect_ids'].cpu().numpy().view(np.uint64)),
      'flat eq',np.array_equal(st['bins']['flatten_ids'],meta['flatten_ids'].cpu().numpy()),
      'offs eq',np.array_equal(st['bins']['isect_offsets'],meta['isect_offsets'][0].cpu().numpy()))
    fw=st['fwd']
    print(' fwd err render',err(r[0].detach().cpu(),st['render']),'alpha',err(a[0].detach().cpu(),fw['alpha']),'ed',err(ed[0].detach().cpu(),fw['exp_depth']),'md',err(md[0].detach().cpu(),fw['med_depth']),'n',err(n[0].detach().cpu(),fw['normal']))
    print(' last eq frac',float((meta['last_ids'][0].cpu().numpy()==fw['last_ids']).mean()),'med eq frac',float((meta['median_ids'][0].cpu().numpy()==fw['median_ids']).mean()))
    g=torch.Generator().manual_seed(7)
    ws=[torch.rand(t.shape,generator=g) for t in (r,a,ed,md,n)]
    loss=sum((t*w.to(dev)).sum() for t,w in zip((r,a,ed,md,n),ws))
    meta['means2d'].retain_grad()
    loss.backward(); torch.cuda.synchronize()
    gr=cr.backward(st,*[w[0].numpy() for w in ws])
    sg=torch.sigmoid(sc['opacity_logits']).numpy()
    print(' grad err means',err(means.grad.cpu(),gr['v_means']),'quats',err(quats.grad.cpu(),gr['v_quats']),
      'logscales',err(ls.grad.cpu(),gr['v_scales']*scales),'opl',err(ol.grad.cpu(),gr['v_opacities']*sg*(1-sg)),'sh',err(sh.grad.cpu(),gr['v_colors']),
      'means2d',err(meta['means2d'].grad[0].cpu(),gr['v_means2d']),'abs',err(meta['means2d'].absgrad[0].cpu(),gr['v_means2d_abs']))
