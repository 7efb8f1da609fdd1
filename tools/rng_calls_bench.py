"""Development aid: ns per call of the scalar draws of the numpy-compatible stream (gammas of the shapes a fit uses)."""
import os, sys, time
ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, ROOT)
import numpy as np
from fokl_gpy_amd import _capi
np.random.seed(1); st = _capi.LegacyStream()
for shape in (5e5 + 30, 34.0, 4.5, 1.0, 0.7):
    st.gammas(shape, 1.0, 1000)
    t = time.perf_counter(); st.gammas(shape, 1.0, 2_000_000); dt = time.perf_counter() - t
    print(f'std_gamma(shape = {shape:g}): {dt / 2e6 * 1e9:6.1f} ns per draw')
t = time.perf_counter(); x = st.normals(20_000_000); dt = time.perf_counter() - t
print(f'normals, finished (blocks of 256): {dt / 2e7 * 1e9:6.2f} ns per normal')
t = time.perf_counter(); x = np.random.standard_normal(20_000_000); dt = time.perf_counter() - t
print(f'numpy legacy standard_normal: {dt / 2e7 * 1e9:6.2f} ns per normal')
t = time.perf_counter(); x = np.random.gamma(5e5, 1.0, 2_000_000); dt = time.perf_counter() - t
print(f'numpy legacy gamma(5e5): {dt / 2e6 * 1e9:6.1f} ns per draw')
