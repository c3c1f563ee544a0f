import os, sys; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch, numpy as np
from tests.closed_loop import fly_engine, fly_oracle
from tests.scenarios import afa
from oracle import oracle_py as ora
b, lo = fly_oracle(ora, 2, 10.0)
names=['pos']*3+['vel']*3+['att']*4+['w']*3
for prec in (afa.AFE_F64, afa.AFE_F32):
    e, le = fly_engine(afa, 2, 10.0, prec)
    d = np.abs(le-lo)/np.maximum(np.abs(lo),1.0)
    per = d.max(axis=(0,2))
    print("precision",prec, "worst", d.max(), {n+str(i):float(per[i]) for i,n in enumerate(names)})
    when = np.argmax(d.max(axis=(1,2))); print(" worst at log index", when)
    for T in (100,300,999): print("  t=%d0ms worst %.3g"%(T, d[:T].max()))
    e.close()
