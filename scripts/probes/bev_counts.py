import numpy as np, torch, sys
sys.path.insert(0,'/root/repo')
import heterofusionrcnn_amd as hf
from bench import rand_bev
rng = np.random.default_rng(3)
a = torch.from_numpy(rand_bev(rng, 70000)).cuda(); g = torch.from_numpy(rand_bev(rng, 64)).cuda()
ov, iou = hf.compute_bev_iou(a, g)
nz = (ov > 0).sum().item()
print("nonzero overlaps", nz, "per 80-row tile", nz / (70000/80))
per = (ov > 0).view(875, 80, 64).sum((1,2)).float()
print("per tile: mean %.1f max %d p90 %.0f" % (per.mean().item(), per.max().item(), per.quantile(0.9).item()))
an = a.cpu().numpy(); gn = g.cpu().numpy()
ca = np.stack([(an[:,0]+an[:,2])/2,(an[:,1]+an[:,3])/2],1); cg = np.stack([(gn[:,0]+gn[:,2])/2,(gn[:,1]+gn[:,3])/2],1)
ra = (abs(an[:,2]-an[:,0])+abs(an[:,3]-an[:,1]))/2; rg=(abs(gn[:,2]-gn[:,0])+abs(gn[:,3]-gn[:,1]))/2
d = np.linalg.norm(ca[:,None]-cg[None],axis=-1)
print("circle survivors", (d <= ra[:,None]+rg[None]).sum(), "per tile", (d <= ra[:,None]+rg[None]).sum()/875)
print("box sizes a", np.abs(an[:,2]-an[:,0]).mean(), np.abs(an[:,3]-an[:,1]).mean(), "extent", ca.min(0), ca.max(0))
