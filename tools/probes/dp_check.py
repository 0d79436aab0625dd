import numpy as np, torch, sys, os
sys.path.insert(0, os.getcwd())
from nanosnp_amd import _lib
from nanosnp_amd.fixtures import load_pileup_weights, seeded_hap_weights
from tests.helpers import golden
ctx=_lib.Context(0); ctx.pileup_load_weights(load_pileup_weights())
z=np.load(golden("pileup_fwd.npz"))
x=torch.from_numpy(z["x"].astype(np.int32)).cuda()
for prec in (0,1):
    ctx.set_option("pileup_precision",prec)
    gt,zy=ctx.pileup_forward(x); torch.cuda.synchronize()
    print("pileup prec",prec,"max|dp|", max(np.abs(gt.cpu().numpy()-z["gt"]).max(), np.abs(zy.cpu().numpy()-z["zy"]).max()))
z=np.load(golden("hap_fwd_large.npz"))
c=_lib.Context(0); c.hap_load_weights(seeded_hap_weights(int(z["seed"]),H=256,ih_scale=0.03,head_scale=120.0))
xs=[]
for t in ("p","h"):
    pl=[torch.from_numpy(z[f"{t}_{k}"].astype(np.int32)).cuda() for k in ("seq","bq","mq","hap")]
    xs.append(c.hap_features(*pl, torch.from_numpy(z[f"{t}_ref"].astype(np.int32)).cuda()))
for prec in (0,1):
    c.set_option("hap_precision",prec)
    gt,zy=c.hap_forward(xs[0],xs[1]); torch.cuda.synchronize()
    print("hap large prec",prec,"max|dp|", max(np.abs(gt.cpu().numpy()-z["gt"]).max(), np.abs(zy.cpu().numpy()-z["zy"]).max()))
for name,seed,kw in (("hap_fwd_h256.npz",12,{}),("hap_fwd_h256x.npz",13,{"ih_scale":0.03,"head_scale":120.0})):
    z=np.load(golden(name)); c2=_lib.Context(0); c2.hap_load_weights(seeded_hap_weights(seed,H=256,**kw))
    gt,zy=c2.hap_forward(torch.from_numpy(z["xp"]).cuda(), torch.from_numpy(z["xh"]).cuda()); torch.cuda.synchronize()
    print(name,"fp32 max|dp|", max(np.abs(gt.cpu().numpy()-z["gt"]).max(), np.abs(zy.cpu().numpy()-z["zy"]).max()))
