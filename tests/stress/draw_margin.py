"""Where the draws of the benchmarked fit (configs[2]) sit relative to the oracle's golden, and why.

    FOKL_FINISH_LOG=exact|fast python tests/stress/draw_margin.py [golden name]

Prints, for the fit of tests/golden/<name>.npz's workload on the GPU: the largest |draw - golden| / column scale
(the quantity bench.py and tests/test_config_goldens.py bound by 1e-9), where it sits, how the error is spread over the
columns, and the conditioning of the final model's eigenproblem -- an eigenvector of XtX moves by about
eps * ||XtX|| / gap under a rounding-level change of XtX (GPU Gram vs the oracle's BLAS Gram), and betas = w Q'.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

import test_config_goldens as T  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else 'cfg2_n1e6_m8'
    g = T.load_golden(name)
    model, betas, mtx, evs, state = T.fit_like_golden(g)
    gb = g['betas']
    scale = np.max(np.abs(gb), axis=0)
    err = np.abs(betas - gb) / scale
    k, j = np.unravel_index(np.argmax(err), err.shape)
    per_col = np.max(err, axis=0)
    out = dict(mode=os.environ.get('FOKL_FINISH_LOG', 'fast (default)'), golden=name,
               max_draw_err_over_scale=float(err.max()), at_draw=int(k), at_column=int(j),
               column_scale=float(scale[j]), per_column_max_quantiles=[float(q) for q in
                                                                       np.quantile(per_col, [0, 0.25, 0.5, 0.75, 1])],
               mean_err_over_scale=float(err.mean()),
               mean_draw_bias_over_scale=float(np.max(np.abs(np.mean(betas - gb, axis=0)) / scale)))
    # conditioning of the final model's eigenproblem, from the fit's own Gram if the model kept it
    stats = getattr(model, 'fit_stats', {})
    out['fit_stats'] = {k: v for k, v in stats.items() if isinstance(v, (int, float))}
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
