import sys; sys.path.insert(0,'/root/repo')
import numpy as np, advancedvi_jl_amd as avi
from oracle import oracle as O
from tests.helpers import SEED, make_family, make_problem
for d,M,ent,count in ((160,160,0,3),(288,224,2,5),(992,352,1,4),(1024,160,0,20)):
    rng=np.random.default_rng(d+M)
    q,_=make_family(rng,d,avi.FULLRANK,np.float32,mu_scale=0.5)
    prob,tgt=make_problem(rng,"diag",d,np.float32)
    params,_=avi.destructure(q)
    ctx=avi.MiviContext(np.float32,avi.FULLRANK,d,M,ent,SEED); ctx.set_problem(prob)
    ref=avi.MiviContext(np.float32,avi.FULLRANK,d,M,ent,SEED); ref.set_problem(prob)
    p=ctx.to_device(params); pr=ref.to_device(params)
    print(d,M,ent,"engine:",bool(ctx.batch_takes_engine(p)),flush=True)
    vals,grads=ctx.estimate_gradient_each(p,11,count); ctx.synchronize()
    vals,grads=vals.cpu().numpy(),grads.cpu().numpy()
    for i in range(count):
        v1,g1=ref.estimate_gradient(pr,11+i)
        _,eps=ref.sample(pr,11+i)
        o=O.estimate_gradient(params.astype(np.float64),d,avi.FULLRANK,tgt,eps.cpu().numpy().astype(np.float64),ent)
        print("  est",i,"value rel vs single %.2e vs oracle %.2e | grad rel vs single %.2e vs oracle %.2e"%(abs(vals[i]-v1.item())/abs(v1.item()),abs(vals[i]-o["value"])/abs(o["value"]),
              np.linalg.norm(grads[i]-g1.cpu().numpy())/np.linalg.norm(g1.cpu().numpy()), np.linalg.norm(grads[i]-o["grad"])/np.linalg.norm(o["grad"])),
              "upper zeros", not np.any(np.triu(grads[i][d:].reshape(d,d,order="F"),1)), "nan", np.isnan(grads[i]).any(),flush=True)
