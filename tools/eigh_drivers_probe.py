"""LAPACK's dsyevr (what scipy.linalg.eigh calls, FR:1499), dsyevd (divide and conquer) and dsyev (QR) on one host thread by model
size: microseconds, deviation of the chain's noise map from an 80-bit Jacobi reference, and dsyevd against dsyevr (development aid;
profiles/eigh_drivers_r04.txt).  The pool uses dsyevd from FOKL_EIGH_DC_FROM columns on (default 80)."""
import os, sys, time
os.environ['OPENBLAS_NUM_THREADS']='1'
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, scipy.linalg as sl
from eigh_device_probe import gram_like, jacobi_truth, canonical, draw_map
rng=np.random.default_rng(3)
print(f"{'n':>4} {'evr us':>8} {'evd us':>8} {'ev us':>8} {'evr-truth':>10} {'evd-truth':>10} {'evd-evr':>10}")
for n in (32, 48, 66, 80, 96, 112, 128, 144):
    g=gram_like(n+1,rng); A=g[:n,:n]
    t={}
    for drv in ('evr','evd','ev'):
        sl.eigh(A,driver=drv)
        t0=time.perf_counter()
        for _ in range(20): l,Q=sl.eigh(A,driver=drv)
        t[drv]=(time.perf_counter()-t0)/20*1e6
    lr,Qr=sl.eigh(A,driver='evr'); ld,Qd=sl.eigh(A,driver='evd')
    tl,tQ=jacobi_truth(A); T=draw_map(tl,canonical(tQ)); sc=np.abs(T).max()
    print(f"{n:4d} {t['evr']:8.1f} {t['evd']:8.1f} {t['ev']:8.1f} {np.abs(draw_map(lr,canonical(Qr))-T).max()/sc:10.2e} {np.abs(draw_map(ld,canonical(Qd))-T).max()/sc:10.2e} {np.abs(draw_map(ld,canonical(Qd))-draw_map(lr,canonical(Qr))).max()/sc:10.2e}")
