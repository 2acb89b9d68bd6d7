import numpy as np, sys
sys.path.insert(0,'.')
import advancedvi_jl_amd as avi
from oracle import oracle as O
from tests.helpers import *
for family in (0,1):
  for d,M in ((40,24),(1024,256)):
    for dtype in (np.float32,):
      for ent in range(5):
        rng=np.random.default_rng(1234+d+7*M)
        q,q_o=make_family(rng,d,family,dtype)
        prob,tgt=make_problem(rng,"diag",d,dtype)
        params,_=avi.destructure(q)
        ctx=avi.MiviContext(dtype,family,d,M,ent,SEED); ctx.set_problem(prob)
        Z,eps=ctx.sample(params,3); eps=eps.cpu().numpy().astype(np.float64)
        ref=O.estimate_gradient(O.destructure(q_o),d,family,tgt,eps,ent)
        vals=[]
        for rep in range(4):
            v,g=ctx.estimate_gradient(params,3); vals.append(float(v.item()))
        g=g.cpu().numpy()
        print(family,d,M,ent,"ref",ref['value'],"got",vals,"grad relerr",rel_err(g,ref['grad']), flush=True)
        ctx.close()
