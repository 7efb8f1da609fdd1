"""Time of the Gibbs recursion on a finished tape (2000 iterations) by model size; FOKL_CHAIN_ISA=base|avx2|avx512 picks the
vector statement (development aid)."""
import sys, time, numpy as np
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from fokl_gpy_amd import _capi
np.random.seed(5)
for p1 in (20, 60, 100, 144):
    s=_capi.LegacyStream()
    tape=_capi.noise_tape(p1, 2000, 5e5+p1/2, 3+p1/2, s)
    _capi.finish_tape_blocks(tape)
    lamb=np.sort(np.random.rand(p1)*1e5+10); qty=np.random.randn(p1)*100
    best=1e9
    for r in range(20):
        t=time.perf_counter(); w,f=_capi.gibbs_chain_from_finished_tape(lamb,qty,900.0,2.0,5e5,0.3,0.9,tape); best=min(best,time.perf_counter()-t)
    print(p1, round(best*1e6,1),'us', round(best/2000*1e9,1),'ns/iter', float(w[-1,0]), float(w.sum()))
