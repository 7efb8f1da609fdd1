"""
GP_Integrate -- Runge-Kutta integration of a dynamical system whose right-hand sides are fitted BSS-ANOVA models.

Same call as the reference's ``FoKL.GP_Integrate.GP_Integrate`` (/root/reference/src/FoKL/GP_Integrate.py:5-282,
"GI" below); the loop itself runs in libfokl_hip.so (``fokl_gp_integrate``, csrc/fokl_integrate.cpp): thousands of
dependent steps with a few hundred flops each are host work by nature.

Behaviour kept from the reference:
  * cubic-spline models only, evaluated on 498 intervals (GI:103-131) although the table has 499 pieces;
  * ``y0`` is advanced in place (GI:271 ``y += ...``) and ``Y[:, 0]`` is the initial state;
  * how a model's input vector is assembled (GI:174-199): its used states in order, normalised with ``norms`` and
    clamped to [0, 1]; then the forcing values -- the FIRST used forcing input contributes the whole row ``b[t]``,
    every further one appends its own column; a model reads only its first ``mtx.shape[1]`` inputs;
  * re-ordering through ``used_inputs`` entries > 1 does not work in the reference (GI:62-67 builds a one-element
    array and indexes past it): an ``IndexError`` is raised here as well.
Not kept: the hard-wired ``np.reshape(y, [2, 1])`` of GI:272 -- any number of states works.
"""
import ctypes

import numpy as np

from . import _capi
from . import getKernels


def GP_Integrate(betas, matrix, b, norms, phis, start, stop, y0, h, used_inputs):
    """
    betas       : list of 1-D coefficient vectors (constant first), one per integrated state -- a draw or the mean
    matrix      : list of interaction matrices, one per state
    b           : forcing inputs over the integration period, already normalised: [steps] or [steps, n_other]
    norms       : [2, n_states] minima (row 0) and maxima (row 1) of the integrated states in the training data
    phis        : cubic-spline coefficients (``model.phis``)
    start, stop, h : T = np.arange(start, stop + h, h)
    y0          : initial state [n_states]; advanced in place
    used_inputs : per state, one flag per (state..., forcing...) input of its model
    returns (T, Y) with Y [n_states, len(T)]
    """
    n_states = len(y0)
    if len(betas) != n_states or len(matrix) != n_states or len(used_inputs) != n_states:
        raise ValueError("betas, matrix and used_inputs need one entry per integrated state")
    T = np.arange(start, stop + h, h)
    n_steps = len(T) - 1
    b = np.asarray(b, dtype=np.float64)
    n_other = int(b.size / b.shape[0]) if b.size > 0 else 0           # GI:179
    forcing = np.ascontiguousarray(b.reshape(b.shape[0], n_other)) if n_other else np.zeros((0, 0))
    if n_other and forcing.shape[0] < n_steps:
        raise IndexError(f"b has {forcing.shape[0]} rows but {n_steps} steps are integrated")   # GI:187 b[ind - 1]
    norms = np.ascontiguousarray(norms, dtype=np.float64)
    if norms.shape != (2, n_states):
        raise ValueError("norms must be [2, n_states]: minima on top of maxima")

    packed, n_basis, width = getKernels.pack_phis(phis, 0)
    sources, coeffs, orders = [], [], []
    for k in range(n_states):
        used = np.asarray(used_inputs[k])
        if used.shape[0] < n_states + n_other:
            raise IndexError("used_inputs entries need one flag per state and per forcing input")
        if np.amax(used) > 1:
            raise IndexError("re-ordering inputs through used_inputs > 1 fails in the reference (GP_Integrate.py:62-67)")
        src = [j for j in range(n_states) if used[j] != 0]            # GI:174-178
        first = True
        for jj in range(n_states, n_states + n_other):                # GI:181-196
            if used[jj] != 0:
                src.extend(-(c + 1) for c in range(n_other)) if first else src.append(-(jj - n_states + 1))
                first = False
        mtx = np.ascontiguousarray(np.atleast_2d(matrix[k]), dtype=np.int32)
        beta = np.ascontiguousarray(np.reshape(betas[k], -1), dtype=np.float64)
        if beta.shape[0] != mtx.shape[0] + 1:
            raise ValueError("every coefficient vector needs one entry per row of its matrix plus the constant")
        if len(src) < mtx.shape[1]:
            raise IndexError("a model has more input columns than used_inputs routes to it")   # GI:125 x[j]
        sources.append(np.array(src, dtype=np.int32))
        coeffs.append(beta)
        orders.append(mtx)

    def pointer_array(arrays):
        return (ctypes.c_void_p * len(arrays))(*[_capi._ptr(a) for a in arrays])

    y = np.asarray(y0)
    state = np.ascontiguousarray(y, dtype=np.float64).copy()
    Y = np.empty((n_states, n_steps + 1), dtype=np.float64)
    rows = np.array([m.shape[0] for m in orders], dtype=np.int32)
    cols = np.array([m.shape[1] for m in orders], dtype=np.int32)
    n_src = np.array([s.shape[0] for s in sources], dtype=np.int32)
    lib = _capi.load()
    _capi._check(lib.fokl_gp_integrate(n_states, n_other, n_steps, pointer_array(coeffs), pointer_array(orders),
                                       _capi._ptr(rows), _capi._ptr(cols), pointer_array(sources), _capi._ptr(n_src),
                                       _capi._ptr(forcing) if n_other else None, _capi._ptr(norms),
                                       _capi._ptr(packed), int(n_basis), int(width), float(h), _capi._ptr(state),
                                       _capi._ptr(Y)))
    try:
        y[...] = state                                                # the reference advances y0 in place (GI:271)
    except (TypeError, ValueError):
        pass
    return T, Y
