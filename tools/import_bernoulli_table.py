"""
One-off import of the reference's Bernoulli coefficient table (DATA, not code).

    python tools/import_bernoulli_table.py /root/reference/src/FoKL/kernels/orthogonal_Bn_scaled.txt

Reads the 20 x 21 whitespace-separated fp64 table (row n = scaled orthonormal Bernoulli polynomial of
order n + 1, column k = coefficient of x**k; ref GK:308-326) and stores the identical fp64 values as
fokl_gpy_amd/kernels/bernoulli_bn_scaled.npy.  The values are printed with 18 significant digits in the
source file, so the float64 round trip is exact.
"""
import sys, os
import numpy as np

src = sys.argv[1]
tab = np.loadtxt(src, delimiter=' ', dtype=np.double)
assert tab.shape == (20, 21), tab.shape
dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'fokl_gpy_amd', 'kernels', 'bernoulli_bn_scaled.npy')
np.save(dst, tab)
print("wrote", os.path.normpath(dst), tab.shape, "max|c| =", np.abs(tab).max())
