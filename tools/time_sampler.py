"""Host sampler micro-timing (development aid): ns per normal for the raw stream and for whole chains."""
import os, sys, time
ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, ROOT)
import numpy as np
from fokl_gpy_amd import _capi
np.random.seed(1); st = _capi.LegacyStream()
x = st.normals(10_000_000)
for rep in range(3):
    t = time.time(); x = st.normals(10_000_000); dt = time.time() - t; print('raw ns/normal', dt / 1e7 * 1e9)
for p in (30, 60, 150):
    lamb = np.linspace(1, 1e6, p); qty = np.random.default_rng(0).standard_normal(p) * 100
    t = time.time()
    for _ in range(10): w = _capi.gibbs_chain(lamb, qty, 5e5, 30.0, 2500.0, 1.0, 1e6, 0.5, 1.0, 2000, st)
    dt = (time.time() - t) / 10
    t = time.time()
    for _ in range(10): tape = _capi.noise_tape(p, 2000, 5e5, 30.0, st)
    dt2 = (time.time() - t) / 10
    t = time.time()
    for _ in range(10): w, neg = _capi.gibbs_chain_from_tape(lamb, qty, 2500.0, 1.0, 1e6, 0.5, 1.0, tape)
    dt3 = (time.time() - t) / 10
    print('P', p, 'chain ms', round(dt * 1e3, 3), 'ns/normal', round(dt / (2000 * p) * 1e9, 2), '| tape ms', round(dt2 * 1e3, 3), 'ns/normal', round(dt2 / (2000 * p) * 1e9, 2), '| arithmetic ms', round(dt3 * 1e3, 3))
