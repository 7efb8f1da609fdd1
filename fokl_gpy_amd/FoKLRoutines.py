"""
``FoKLRoutines.FoKL`` -- drop-in class surface of FoKL-GPy for the forward-selection hot path, MI355X-native.

Keeps the reference's public surface (``src/FoKL/FoKLRoutines.py``, "FR"): the constructor keywords and
defaults (FR:205-216), ``clean`` (FR:441), ``fit`` (FR:1202), ``evaluate`` (FR:851), ``coverage3`` (FR:982),
``bss_derivatives`` (FR:594), ``evaluate_basis`` (FR:807), ``save`` / ``load`` / ``clear`` and the result attributes
``betas, avg_betas, mtx, evs, inputs, data, minmax, trainlog``.  The numerics of ``fit``, ``evaluate`` and
``bss_derivatives`` run on the GPU through ``libfokl_hip.so`` (see ``engine.py`` and ``include/fokl_hip.h``); there
is no CPU fallback -- without the library or a gfx950 device they raise.

Deliberately NOT rebuilt (out of the hot-path scope): ``fitupdate`` (``update=True``),
``to_pyomo``; calling them raises ``NotImplementedError``.

Device selection is by environment (``FOKL_DEVICE``, else ``LOCAL_RANK``, else 0), never by a new keyword:
unknown keywords must keep raising ``ValueError`` exactly like the reference (FR:78).
"""
import copy
import os
import pickle
import sys
import threading
import time
import warnings

import numpy as np

from . import getKernels
from . import _capi
from . import engine as _engine
from . import update as _update


def _column_min_max(a):
    """(min, max) of every column of a 2-D array -- np.min / np.max per column (exact, NaN-propagating), computed over
    rows of 64 x m values so that the reduction runs along a long contiguous axis (N = 1e6, M = 8: 49 -> 7 ms)."""
    n, m = a.shape
    if a.flags.c_contiguous and a.dtype == np.float64 and n * m >= 1 << 18:
        try:
            return _capi.column_min_max(a)          # the same minima / maxima on several native threads
        except _capi.FoklNativeError:
            pass
    group = 64
    whole = (n // group) * group
    if not a.flags.c_contiguous or whole == 0:
        return np.min(a, axis=0), np.max(a, axis=0)
    wide = a[:whole].reshape(n // group, group * m)
    lows = wide.min(axis=0).reshape(group, m).min(axis=0)
    highs = wide.max(axis=0).reshape(group, m).max(axis=0)
    if whole < n:
        lows = np.minimum(lows, a[whole:].min(axis=0))
        highs = np.maximum(highs, a[whole:].max(axis=0))
    return lows, highs


class _ModelUnpickler(pickle.Unpickler):
    """``.fokl`` files are pickles of the model object (FR:1840); files written by the reference name its class
    ``FoKL.FoKLRoutines.FoKL`` (or ``src.FoKL...`` from a source checkout) -- both load as this package's class, whose
    attribute names are the reference's."""

    def find_class(self, module, name):
        if name == 'FoKL' and module.split('.')[-2:] == ['FoKL', 'FoKLRoutines']:
            return FoKL
        return super().find_class(module, name)


def load(filename, directory=None):
    """Load a model written by ``FoKL.save`` -- this package's or the reference's (same contract as FR:24-46)."""
    if filename[-5::] != ".fokl":
        filename = filename + ".fokl"
    path = os.path.join(directory, filename) if directory is not None else filename
    with open(path, "rb") as fh:
        return _ModelUnpickler(fh).load()


_TRUE_WORDS = ('yes', 'y', 'on', 'all', 'true', 'both')
_FALSE_WORDS = ('no', 'n', 'off', 'none', 'n/a', 'false')


def _str_to_bool(s):
    """'on'/'off'-style strings, None, [] and numbers -> bool (behaviour of FR:49-68)."""
    if isinstance(s, str):
        if s in _TRUE_WORDS:
            return True
        if s in _FALSE_WORDS:
            return False
        warnings.warn(f"Could not understand string '{s}' as a boolean.", category=UserWarning)
        return s
    if s is None or not s:
        return False
    try:
        return bool(s != 0)
    except Exception:
        warnings.warn("Could not convert non-string to a boolean.", category=UserWarning)
        return s


def _process_kwargs(default, user):
    """Merge user keywords into ``default`` (dict) or only validate them against it (list); FR:71-90."""
    if isinstance(default, dict):
        if not isinstance(user, dict):
            raise ValueError("Input 'user' must be a dictionary formed by kwargs.")
        for kw, val in user.items():
            if kw not in default:
                raise ValueError(f"Unexpected keyword argument: '{kw}'")
            default[kw] = val
        return default
    if isinstance(default, list):
        for kw in user.keys():
            if kw not in default:
                raise ValueError(f"Unexpected keyword argument: '{kw}'")
        return user
    raise ValueError("Input 'default' must be a dictionary or list.")


def _set_attributes(self, attrs):
    if isinstance(attrs, dict):
        for key, value in attrs.items():
            setattr(self, key, value)
    else:
        warnings.warn("Input must be a Python dictionary.")


def _merge_dicts(d1, d2):
    d = d1.copy()
    d.update(d2)
    return d


# device contexts are process-wide and never stored on the model (models must stay picklable, FR:1807-1846)
_CONTEXTS = {}


def _default_device():
    for var in ('FOKL_DEVICE', 'LOCAL_RANK'):
        if os.environ.get(var, '') != '':
            return int(os.environ[var])
    return 0


def device_backend(device=None):
    """The process-wide HIP backend of ``device`` (created on first use; raises if no gfx950 device)."""
    device = _default_device() if device is None else int(device)
    if device not in _CONTEXTS:
        _CONTEXTS[device] = _engine.HipBackend(_capi.DeviceContext(device))
    return _CONTEXTS[device]


_BLAS_LIMIT_LOCK = threading.Lock()
_BLAS_LIMIT_USERS = 0
_BLAS_LIMIT_CTX = None


_BLAS_CONTROLLER = (None, None)


def _blas_controller():
    """threadpoolctl looks through every loaded shared library each time a limit is taken (1-2 ms per fit, measured
    with FOKL_POOL_TRACE); the libraries do not change between fits, so the controller is kept -- and made again when
    scipy's BLAS has been loaded since."""
    global _BLAS_CONTROLLER
    key = ('scipy.linalg' in sys.modules, 'scipy.linalg.cython_lapack' in sys.modules)
    if _BLAS_CONTROLLER[0] != key:
        from threadpoolctl import ThreadpoolController
        _BLAS_CONTROLLER = (key, ThreadpoolController())
    return _BLAS_CONTROLLER[1]


class _host_blas_threads:
    """The per-candidate host algebra is tiny ((P+1)^2 eigenproblems, draws x (P+1) products): a BLAS pool sized for
    a 256-thread host spends more time waking threads than computing, and its spinning workers compete with the
    noise-tape thread.  Cap it (FOKL_HOST_THREADS, default 1: measured best on a 2 x 64-core EPYC host) for as long as
    any fit of this process is running -- fits may run side by side on threads (bench.py --config 4 --concurrent), so
    the cap is taken by the first and released by the last."""

    def __enter__(self):
        global _BLAS_LIMIT_USERS, _BLAS_LIMIT_CTX
        with _BLAS_LIMIT_LOCK:
            _BLAS_LIMIT_USERS += 1
            if _BLAS_LIMIT_USERS == 1:
                try:
                    _BLAS_LIMIT_CTX = _blas_controller().limit(limits=int(os.environ.get('FOKL_HOST_THREADS', '1')),
                                                               user_api='blas')
                except Exception:
                    _BLAS_LIMIT_CTX = None
        return self

    def __exit__(self, *exc):
        global _BLAS_LIMIT_USERS, _BLAS_LIMIT_CTX
        with _BLAS_LIMIT_LOCK:
            _BLAS_LIMIT_USERS -= 1
            if _BLAS_LIMIT_USERS == 0 and _BLAS_LIMIT_CTX is not None:
                _BLAS_LIMIT_CTX.restore_original_limits()
                _BLAS_LIMIT_CTX = None
        return False


_CLEAN_DEFAULTS = {'train': 1, 'AutoTranspose': True, 'SingleInstance': False, 'bit': 64,
                   'normalize': True, 'minmax': None, 'pillow': None, 'pillow_type': 'percent'}


class _DeviceInputs:
    """The normalised inputs of a model whose fit(clean=True) normalised them on the device (fokl_stage_inputs /
    fokl_upload_staged): they exist there only until somebody reads the model's ``inputs`` attribute, which fetches them
    (fokl_download_inputs: the numbers FoKL.clean produces on the host, bit for bit) and keeps the array."""
    ndim = 2
    dtype = np.dtype(np.float64)

    def __init__(self, backend, shape):
        self.backend, self.shape = backend, tuple(shape)

    def fetch(self):
        return self.backend.download_inputs()


class FoKL:
    # ``inputs`` is stored under its own name in the instance (files written by ``save`` and by the reference carry it
    # there); reading it turns a _DeviceInputs into the array it stands for.
    def _get_inputs(self):
        try:
            value = self.__dict__['inputs']
        except KeyError:
            raise AttributeError("'FoKL' object has no attribute 'inputs'") from None
        if isinstance(value, _DeviceInputs):
            value = self.__dict__['inputs'] = value.fetch()
        return value

    def _set_inputs(self, value):
        self.__dict__['inputs'] = value

    def _del_inputs(self):
        try:
            del self.__dict__['inputs']
        except KeyError:
            raise AttributeError('inputs') from None

    inputs = property(_get_inputs, _set_inputs, _del_inputs)

    def _materialise_inputs(self):
        if isinstance(self.__dict__.get('inputs'), _DeviceInputs):
            self._get_inputs()

    def __getstate__(self):
        self._materialise_inputs()
        return self.__dict__

    def __copy__(self):
        self._materialise_inputs()
        twin = type(self).__new__(type(self))
        twin.__dict__.update(self.__dict__)
        return twin

    def __init__(self, **kwargs):
        """
        Hyper-parameters and defaults as in the reference (FR:168-216):

            kernel='Cubic Splines' | 'Bernoulli Polynomials' (or the index 0 | 1), phis=f(kernel), relats_in=[],
            a=4, b=f(a, data), atau=4, btau=f(atau, data), tolerance=3, burnin=1000, draws=1000, gimmie=False,
            way3=False, threshav=0.05, threshstda=0.5, threshstdb=2, aic=False,
            sigsqd0=0.5, burn=500, update=False, built=False, UserWarnings=True, ConsoleOutput=True
        """
        self.hypers = ['kernel', 'phis', 'relats_in', 'a', 'b', 'atau', 'btau', 'tolerance', 'burnin', 'draws',
                       'gimmie', 'way3', 'threshav', 'threshstda', 'threshstdb', 'aic', 'update', 'built']
        self.settings = ['UserWarnings', 'ConsoleOutput']
        self.kernels = ['Cubic Splines', 'Bernoulli Polynomials']
        self.keep = ['keep', 'hypers', 'settings', 'kernels'] + self.hypers + self.settings + self.kernels

        current = _process_kwargs({
            'kernel': 'Cubic Splines', 'phis': None, 'relats_in': [], 'a': 4, 'b': None, 'atau': 4, 'btau': None,
            'tolerance': 3, 'burnin': 1000, 'draws': 1000, 'gimmie': False, 'way3': False, 'threshav': 0.05,
            'threshstda': 0.5, 'threshstdb': 2, 'aic': False,
            'sigsqd0': 0.5, 'burn': 500, 'update': False, 'built': False,
            'UserWarnings': True, 'ConsoleOutput': True}, kwargs)
        for flag in ('gimmie', 'way3', 'aic', 'UserWarnings', 'ConsoleOutput'):
            if not (current[flag] is False or current[flag] is True):
                current[flag] = _str_to_bool(current[flag])

        if isinstance(current['kernel'], int):
            current['kernel'] = self.kernels[current['kernel']]
        if current['phis'] is None:
            if current['kernel'] == self.kernels[0]:
                current['phis'] = getKernels.sp500()
            elif current['kernel'] == self.kernels[1]:
                current['phis'] = getKernels.bernoulli()
            elif isinstance(current['kernel'], str):
                raise ValueError(f"The user-provided kernel '{current['phis']}' is not supported.")
            else:
                raise ValueError("The user-provided kernel is not supported.")

        warnings.filterwarnings("default" if current['UserWarnings'] else "ignore", category=UserWarning)
        for key, value in current.items():
            setattr(self, key, value)
        self.setnos = None

    # -----------------------------------------------------------------------------------------------------
    # dataset formatting (host side, not on the device path; behaviour of FR:248-542)
    # -----------------------------------------------------------------------------------------------------

    def _format(self, inputs, data=None, AutoTranspose=True, SingleInstance=False, bit=64, _copy_inputs=True):
        """inputs -> [n, m] ndarray, data -> [n, 1] ndarray of the requested float width (FR:248-316)."""
        import pandas as pd
        AutoTranspose = _str_to_bool(AutoTranspose)
        SingleInstance = _str_to_bool(SingleInstance)
        widths = {16: np.float16, 32: np.float32, 64: np.float64}
        if SingleInstance is True:
            AutoTranspose = False
        if bit not in widths:
            warnings.warn(f"Keyword 'bit={bit}' limited to values of 16, 32, or 64. Assuming default value of 64.",
                          category=UserWarning)
            bit = 64
        dtype = widths[bit]

        if isinstance(inputs, (pd.DataFrame, pd.Series)):
            inputs = inputs.to_numpy()
            warnings.warn("'inputs' was auto-converted to numpy. Convert manually for assured accuracy.",
                          category=UserWarning)
        if data is not None and isinstance(data, (pd.DataFrame, pd.Series)):
            data = data.to_numpy()
            warnings.warn("'data' was auto-converted to numpy. Convert manually for assured accuracy.",
                          category=UserWarning)

        inputs = np.array(inputs) if _copy_inputs else np.asarray(inputs)    # (no copy: the caller only reads them)
        if inputs.ndim > 2:
            inputs = np.squeeze(inputs)
        if inputs.dtype != dtype:
            inputs = np.array(inputs, dtype=dtype)
            warnings.warn(f"'inputs' was converted to float{bit}. May require user-confirmation that "
                          f"values did not get corrupted.", category=UserWarning)
        if inputs.ndim == 1:
            inputs = inputs[np.newaxis, :] if SingleInstance is True else inputs[:, np.newaxis]
        if AutoTranspose is True and SingleInstance is False and inputs.shape[1] > inputs.shape[0]:
            inputs = inputs.transpose()
            warnings.warn("'inputs' was transposed. Ignore if more datapoints than input variables, else set "
                          "'AutoTranspose=False' to disable.", category=UserWarning)

        if data is not None:
            data = np.squeeze(np.array(data))
            if data.dtype != dtype:
                data = np.array(data, dtype=dtype)
                warnings.warn(f"'data' was converted to float{bit}. May require user-confirmation that "
                              f"values did not get corrupted.", category=UserWarning)
            if data.ndim == 1:
                data = data[:, np.newaxis]
            else:
                rows, cols = data.shape[0], data.shape[1]
                if (cols != 1 and rows != 1) or (cols == 1 and rows == 1):
                    raise ValueError("Error: 'data' must be a vector.")
                if cols != 1 and rows == 1:
                    data = data.transpose()
                    warnings.warn("'data' was transposed to match FoKL formatting.", category=UserWarning)
        return inputs, data

    def _normalize(self, inputs, minmax=None, pillow=None, pillow_type='percent', _bounds=None):
        """Min-max normalisation of every input column to [0, 1]; updates ``self.minmax`` (FR:318-439).
        _bounds = (column minima, column maxima) found elsewhere (the device): the bookkeeping only -- returns
        (lows, spans) of the normalisation (x - lows) / spans instead of applying it."""
        mm = inputs.shape[1]
        allowed = ['percent', 'absolute']
        if isinstance(pillow_type, str):
            pillow_type = [pillow_type] * mm
        elif isinstance(pillow_type, list) and len(pillow_type) != mm:
            raise ValueError("Input 'pillow_type' must be string or correspond to input variables (i.e., columns of "
                             "'inputs').")
        for pt in pillow_type:
            if pt not in allowed:
                raise ValueError(f"'pillow_type' is limited to {allowed}.")

        use_pillow = pillow is not None
        if pillow is None:
            pillow = 0.0
        if isinstance(pillow, int):
            pillow = float(pillow)
        if isinstance(pillow, float):
            pillow = [[pillow, pillow]] * mm
        elif isinstance(pillow[0], (int, float)):
            flat = list(pillow)
            if len(flat) == 2:
                pillow = [[float(flat[0]), float(flat[1])]]
                if mm * 2 != 1:          # the reference compares the collapsed length 1 against 2 * mm
                    raise ValueError("Input 'pillow' must correspond to input variables (i.e., columns of 'inputs').")
            elif len(flat) != mm * 2:
                raise ValueError("Input 'pillow' must correspond to input variables (i.e., columns of 'inputs').")
            else:
                pillow = [[float(flat[i]), float(flat[i + 1])] for i in range(0, len(flat), 2)]

        def _bad_minmax():
            raise ValueError("Input 'minmax' must correspond to input variables (i.e., columns of 'inputs').")

        if minmax is None:
            if hasattr(self, 'minmax'):
                minmax = self.minmax
            else:
                # one contiguous pass each instead of a strided pass per column and bound (minima / maxima are exact:
                # the same values as the reference's np.min / np.max per column, FR:395)
                lows, highs = _column_min_max(inputs) if _bounds is None else _bounds
                minmax = list([lows[k], highs[k]] for k in range(mm))
        elif isinstance(minmax[0], (int, float)):
            flat = list(minmax)
            if len(flat) == 2:
                minmax = [flat]
                if mm * 2 != 1:
                    _bad_minmax()
            elif len(flat) != mm * 2:
                _bad_minmax()
            else:
                minmax = [[flat[i], flat[i + 1]] for i in range(0, len(flat), 2)]
        elif len(minmax) != mm:
            _bad_minmax()

        if use_pillow:
            widened = []
            for k in range(mm):
                lo, hi = minmax[k][0], minmax[k][1]
                span = hi - lo
                if pillow_type[k] == 'percent':
                    widened.append([lo - span * pillow[k][0], hi + span * pillow[k][1]])
                else:   # 'absolute': choose [min, max] so that lo -> pillow[k][0] and hi -> 1 - pillow[k][1]
                    q, p1 = pillow[k][0], pillow[k][1]
                    new_lo = lo if q == 0 else (lo * (1 - p1) - hi * q) / (1 - p1 - q)
                    if p1 == 0:
                        new_hi = hi
                    elif q == 0:
                        new_hi = (hi - p1 * new_lo) / (1 - p1)
                    else:
                        new_hi = (lo - new_lo) / q + new_lo
                    widened.append([new_lo, new_hi])
            minmax = widened

        if hasattr(self, 'minmax'):
            if any(minmax[k] == self.minmax[k] for k in range(mm)) is False:
                warnings.warn("The model already contains normalization [min, max] bounds, so the currently trained "
                              "model will not be valid for the new bounds requested. Train a new model with these new "
                              "bounds.", category=UserWarning)
        self.minmax = minmax

        if _bounds is not None:
            return (np.array([float(minmax[k][0]) for k in range(mm)], dtype=np.float64),
                    np.array([float(minmax[k][1] - minmax[k][0]) for k in range(mm)], dtype=np.float64))
        if inputs.dtype == np.float64 and inputs.flags.c_contiguous and inputs.flags.writeable:
            # the reference's per-column statement (FR:436-437) over whole rows: the same subtraction and the same
            # division per element, two contiguous passes instead of 2 m strided ones (N = 1e6, M = 8: 0.14 -> 0.04 s)
            lows = np.array([float(minmax[k][0]) for k in range(mm)], dtype=np.float64)
            spans = np.array([float(minmax[k][1] - minmax[k][0]) for k in range(mm)], dtype=np.float64)
            done = False
            if inputs.size >= 1 << 18:
                try:
                    _capi.normalize_columns(inputs, lows, spans)     # the same two operations per element, native threads
                    done = True
                except _capi.FoklNativeError:
                    done = False
            if not done:
                np.subtract(inputs, lows, out=inputs)
                np.divide(inputs, spans, out=inputs)
        else:
            for k in range(mm):
                inputs[:, k] = (inputs[:, k] - minmax[k][0]) / (minmax[k][1] - minmax[k][0])
        return inputs

    def _clean_on_device(self, inputs, data, current, backend):
        """fit(clean=True) of a large float64 dataset: the rows go to the device as they are, minima / maxima and the
        normalisation happen there (fokl_stage_inputs, fokl_upload_staged: the host's numbers bit for bit), the host
        keeps the bookkeeping (minmax, pillow) and, until somebody reads it, no normalised copy."""
        inputs, data = self._format(inputs, data, current['AutoTranspose'], current['SingleInstance'], current['bit'],
                                    _copy_inputs=False)
        bounds = backend.stage_inputs(inputs, self)
        lows, spans = self._normalize(inputs, current['minmax'], current['pillow'], current['pillow_type'], _bounds=bounds)
        self._staged_upload = (backend, lows, spans, inputs.shape)
        trainlog = self.generate_trainlog(current['train'], inputs.shape[0])
        _set_attributes(self, {'inputs': _DeviceInputs(backend, inputs.shape), 'data': data, 'trainlog': trainlog})
        return self.__dict__['inputs'], data

    def _device_clean_wanted(self, inputs, data, current):
        """The device route covers the plain case: float64 rows [n, m] in C order, n >= m, large enough to matter, all rows
        used for training, 64-bit, the Bernoulli kernel (the spline kernel validates the normalised values on the host)."""
        return (os.environ.get('FOKL_CLEAN', 'device') != 'host' and current['normalize'] is True and
                isinstance(inputs, np.ndarray) and inputs.dtype == np.float64 and inputs.ndim == 2 and
                inputs.flags.c_contiguous and inputs.shape[0] >= inputs.shape[1] and inputs.size >= 1 << 18 and
                data is not None and current['bit'] == 64 and current['train'] == 1 and
                _str_to_bool(current['SingleInstance']) is not True and
                self._kernel_id() == getKernels.KERNEL_BERNOULLI)

    def clean(self, inputs, data=None, kwargs_from_other=None, _setattr=False, _device=None, **kwargs):
        """Format and normalise a dataset; defines ``inputs, data, trainlog`` on first use (FR:441-507).
        _device (fit's own call only): a callable returning the backend the dataset is about to be uploaded to."""
        if kwargs_from_other is not None:
            kwargs = _merge_dicts(kwargs, kwargs_from_other)
        current = _process_kwargs(dict(_CLEAN_DEFAULTS), kwargs)
        current['normalize'] = _str_to_bool(current['normalize'])

        self.__dict__.pop('_staged_upload', None)
        if _device is not None and _setattr is True and self._device_clean_wanted(inputs, data, current):
            backend = _device()
            if hasattr(backend, 'stage_inputs'):
                return self._clean_on_device(inputs, data, current, backend)

        inputs, data = self._format(inputs, data, current['AutoTranspose'], current['SingleInstance'], current['bit'])
        if current['normalize'] is True:
            inputs = self._normalize(inputs, current['minmax'], current['pillow'], current['pillow_type'])
            # NB the reference tests `np.max(mask) is True` (FR:488), which no numpy bool satisfies, so
            # out-of-range values are neither capped nor reported; kept as is.

        if hasattr(self, 'inputs') is False or _setattr is True:
            trainlog = self.generate_trainlog(current['train'], inputs.shape[0])
            _set_attributes(self, {'inputs': inputs, 'data': data, 'trainlog': trainlog})

        if data is None:
            return inputs
        return inputs, data

    def generate_trainlog(self, train, n=None):
        """Random logical vector selecting ``train`` (fraction) of ``n`` rows, or None for all rows (FR:509-530)."""
        if train < 1:
            if n is None:
                n = self.inputs.shape[0]
            want = max(int(n * train), 2)
            picked = np.array([], dtype=int)
            while len(picked) < want:
                with warnings.catch_warnings():
                    warnings.simplefilter('ignore', DeprecationWarning)
                    picked = np.append(picked, np.random.random_integers(n, size=want) - 1)
                picked = np.unique(picked)
                np.random.shuffle(picked)
            picked = picked[0:want]
            trainlog = np.zeros(n, dtype=bool)
            trainlog[picked] = True
            return trainlog
        return None

    def trainset(self):
        """(train inputs, train data) according to ``self.trainlog`` (FR:532-542)."""
        if self.trainlog is None:
            return self.inputs, self.data
        return self.inputs[self.trainlog, :], self.data[self.trainlog]

    def _inputs_to_phind(self, inputs, phis=None, kernel=None):
        """Spline piece index / local coordinate of normalised inputs, host version for validation (FR:544-592)."""
        kernel = self.kernel if kernel is None else kernel
        phis = self.phis if phis is None else phis
        if kernel == self.kernels[1]:
            warnings.warn("Twice normalization of inputs is not required for the 'Bernoulli Polynomials' kernel",
                          category=UserWarning)
            return inputs, [], []
        l_phis = len(phis[0][0])
        phind = np.array(np.ceil(inputs * l_phis), dtype=np.uint16)
        if phind.ndim == 1:
            phind = phind[:, np.newaxis]
        phind = phind + (phind == 0)
        try:
            inputs.dtype
        except AttributeError:
            raise AttributeError("Inputs must be a numpy array, to process automatically try making clean = True")
        r = 1 / l_phis
        X = (inputs - np.array((phind - 1) * r, dtype=inputs.dtype)) / r
        phind = phind - 1
        xsm = np.array(l_phis * inputs - phind, dtype=inputs.dtype)
        if np.max(phind) > 499 or np.min(phind) < 0:
            raise ValueError('Inputs are not normalized correctly, try calling clean=True within evaluate to '
                             'evaluate with normalization of model training')
        return X, phind, xsm

    # -----------------------------------------------------------------------------------------------------
    # scalar basis evaluation (kept for API compatibility; the device kernel is the hot-path implementation)
    # -----------------------------------------------------------------------------------------------------

    def evaluate_basis(self, c, x, kernel=None, d=0):
        """Value (d=0) or d-th derivative of one basis function at ``x`` given its coefficients (FR:807-849)."""
        if kernel is None:
            kernel = self.kernel
        elif isinstance(kernel, int):
            kernel = self.kernels[kernel]
        if kernel not in self.kernels:
            raise ValueError(f"The kernel {kernel} is not currently supported. Please select from the following: "
                             f"{self.kernels}.")
        if kernel == self.kernels[0]:
            if d == 0:
                basis = c[0] + c[1] * x + c[2] * (x ** 2) + c[3] * (x ** 3)
            elif d == 1:
                basis = c[1] + 2 * c[2] * x + 3 * c[3] * (x ** 2)
            elif d == 2:
                basis = 2 * c[2] + 6 * c[3] * x
        else:
            if d == 0:
                basis = c[0] + sum(c[k] * (x ** k) for k in range(1, len(c)))
            elif d == 1:
                basis = c[1] + sum(k * c[k] * (x ** (k - 1)) for k in range(2, len(c)))
            elif d == 2:
                basis = sum((k - 1) * k * c[k] * (x ** (k - 2)) for k in range(2, len(c)))
        return basis

    # -----------------------------------------------------------------------------------------------------
    # device plumbing
    # -----------------------------------------------------------------------------------------------------

    def _kernel_id(self):
        if self.kernel == self.kernels[0]:
            return getKernels.KERNEL_SPLINES
        if self.kernel == self.kernels[1]:
            return getKernels.KERNEL_BERNOULLI
        raise ValueError(f"The kernel {self.kernel} is not currently supported.")

    def _backend(self):
        """Device backend for this process.  ``_backend_override`` is a hook for tests of the host logic only."""
        override = getattr(self, '_backend_override', None)
        if override is not None:
            return override
        return device_backend()

    def _upload(self, backend, inputs, data):
        kid = self._kernel_id()
        staged = self.__dict__.pop('_staged_upload', None)
        if staged is not None and staged[0] is backend:
            packed, nb, width = getKernels.pack_phis(self.phis, kid)
            backend.upload_staged(np.asarray(data, dtype=np.float64), kid, packed, nb, width, staged[1], staged[2], self)
            return
        if isinstance(inputs, _DeviceInputs):
            raise RuntimeError("the dataset was staged on another device backend than the one the fit uploads to")
        inputs = np.asarray(inputs, dtype=np.float64)
        if kid == getKernels.KERNEL_SPLINES:
            self._inputs_to_phind(inputs)              # range validation with the reference's own expression
        packed, nb, width = getKernels.pack_phis(self.phis, kid)
        backend.upload(inputs, np.asarray(data, dtype=np.float64), kid, packed, nb, width)

    # -----------------------------------------------------------------------------------------------------
    # fit
    # -----------------------------------------------------------------------------------------------------

    def fit(self, inputs=None, data=None, **kwargs):
        """
        Train the model by forward variable selection (FR:1202-1760).  Returns ``(betas, mtx, evs)``:
        the last ``draws`` posterior draws of the best model's coefficients, its interaction matrix and the
        BIC of every sub-stage.  Keywords: any hyper-parameter, ``clean`` (+ the keywords of ``clean``),
        ``ConsoleOutput``.
        """
        backend, n, m = self._prepare_fit(inputs, data, kwargs)
        if self.update == True:  # noqa: E712  (sequential updating, FR:1365-1367)
            self.betas, self.mtx, self.evs = self._update_search(backend, n, m)
            return self.betas, self.mtx, self.evs
        return self._search(backend, n, m)

    def _prepare_fit(self, inputs, data, kwargs):
        """Everything ``fit`` does before the search: keyword triage, cleaning, data-driven defaults, upload."""
        try:
            return self._prepare_fit_steps(inputs, data, kwargs)
        except BaseException:
            # a dataset staged for normalisation on the device whose upload never completed (relats_in rejected, a shape
            # error): the placeholder would describe what is NOT on the device -- a later read of ``inputs`` would fetch
            # another dataset's numbers -- so it goes; the model is left without inputs, as before the call
            self.__dict__.pop('_staged_upload', None)
            if isinstance(self.__dict__.get('inputs'), _DeviceInputs):
                del self.__dict__['inputs']
            raise

    def _prepare_fit_steps(self, inputs, data, kwargs):
        t_begin = time.perf_counter()
        fit_opts = {'ConsoleOutput': _str_to_bool(kwargs.get('ConsoleOutput', self.ConsoleOutput)),
                    'clean': _str_to_bool(kwargs.get('clean', False))}
        clean_defaults = dict(_CLEAN_DEFAULTS)
        kwargs = _process_kwargs(self.hypers + list(fit_opts.keys()) + list(clean_defaults.keys()), kwargs)
        if fit_opts['clean'] is False:
            if any(kw in clean_defaults for kw in kwargs):
                warnings.warn("Keywords for automatic cleaning were defined but clean=False.")
            clean_defaults = {}

        kwargs_to_clean = {}
        for kw, val in kwargs.items():
            if kw in self.hypers:
                setattr(self, kw, _str_to_bool(val) if kw in ('gimmie', 'way3', 'aic') else val)
            elif kw in clean_defaults:
                kwargs_to_clean[kw] = val
        self.ConsoleOutput = fit_opts['ConsoleOutput']

        clean_failed = False
        if fit_opts['clean'] is True:
            try:
                if inputs is None:
                    inputs, _ = self.trainset()
                if data is None:
                    _, data = self.trainset()
            except Exception:
                clean_failed = True
            self.clean(inputs, data, kwargs_from_other=kwargs_to_clean, _setattr=True, _device=self._backend)
        else:
            try:
                if inputs is None:
                    inputs, _ = self.trainset()
                if data is None:
                    _, data = self.trainset()
            except Exception:
                warnings.warn("Keyword 'clean' was set to False but is required prior to or during 'fit'. Assuming "
                              "'clean' is True.", category=UserWarning)
                if inputs is None or data is None:
                    clean_failed = True
                else:
                    fit_opts['clean'] = True
                    self.clean(inputs, data, kwargs_from_other=kwargs_to_clean, _setattr=True)
        if clean_failed:
            raise ValueError("'inputs' and/or 'data' were not provided so 'clean' could not be performed.")

        staged = self.__dict__.get('_staged_upload')
        if staged is not None:
            # normalised on the device: all rows train (trainlog is None), the host holds no normalised copy to hand on
            inputs, data = self.__dict__['inputs'], self.data
        else:
            try:
                inputs, data = self.trainset()
            except Exception:
                warnings.warn("If not calling 'clean' prior to 'fit' or within the argument of 'fit', then this is the "
                              "likely source of any subsequent errors. To troubleshoot, simply include 'clean=True' within "
                              "the argument of 'fit'.", category=UserWarning)

            self.inputs = inputs
            self.data = data


        # data-driven defaults of the inverse-gamma scales (FR:1322-1348)
        a, atau = self.a, self.atau
        b, btau = self.b, self.btau
        if btau is None or b is None:
            data64 = np.asarray(data, dtype=np.float64)
            if data.dtype != np.float64:
                data_mean = np.sum(data64) / data64.shape[0]
                sigmasq = np.sum((data64 - data_mean) ** 2) / (data64.shape[0] - 1)
            else:
                sigmasq = np.var(data)
                data_mean = np.mean(data)
            if sigmasq == np.inf:
                warnings.warn("The dataset is too large such that 'sigmasq=inf' even as 64-bit. Consider training on "
                              "a smaller percentage of the dataset.", category=UserWarning)
            if b is None:
                b = sigmasq * (a + 1)
                self.b = b
            if btau is None:
                btau = (np.abs(data_mean) / sigmasq) * (atau + 1)
                self.btau = btau

        if self.update != True:  # noqa: E712  (fitupdate has its own handling of relats_in, FR:2449-2468)
            self._check_relats(np.shape(inputs)[1])

        backend = self._backend()
        t_upload = time.perf_counter()
        self._upload(backend, inputs, data)
        sync = getattr(getattr(backend, 'ctx', None), 'sync', None)
        if sync is not None:
            sync()                                   # H2D copy + transposition to structure-of-arrays have completed
        t_done = time.perf_counter()
        # what a fit costs before the search starts: formatting / normalisation / defaults on the host, then the
        # upload (reported by bench.py next to the search's own time)
        self.prepare_stats = dict(clean_s=t_upload - t_begin, upload_s=t_done - t_upload)
        return backend, np.shape(inputs)[0], np.shape(inputs)[1]

    def _search(self, backend, n, m, n_global=None, row_sharded=False, comm=None, candidate_sharded=False,
                rng_state=None):
        """Forward selection on the dataset currently resident on ``backend`` (the timed region of bench.py).
        ``comm`` + ``candidate_sharded``: every rank of the communicator calls this on the same dataset with the same
        numpy stream; candidate models are dealt over the ranks (engine.ForwardSelection).
        ``rng_state``: a numpy legacy state (``np.random.RandomState(seed).get_state()``) to run the chain from instead
        of numpy's global generator -- fits running side by side on threads cannot share the global one; the state
        after the fit is left in ``self._rng_state_after`` and the global generator is not touched."""
        _engine._mark('search_begin')
        stream = _capi.LegacyStream(rng_state)
        state_at_start = stream.as_numpy_state()

        def new_search(stream, device_chains=True):
            search = _engine.ForwardSelection(
                backend, n, m, len(self.phis), self.a, self.b, self.atau, self.btau, self.tolerance,
                self.burnin + self.draws, self.draws, self.gimmie, self.way3, self.threshav, self.threshstda,
                self.threshstdb, self.aic, stream, console=self.ConsoleOutput,
                comm=comm if comm is not None else getattr(self, '_comm', None),
                row_sharded=row_sharded, n_global=n_global, candidate_sharded=candidate_sharded)
            search.allow_device_chains = device_chains
            search.allow_direct_decisions = device_chains        # (a repeated search decides every kill test from its own G2)
            return search

        search = new_search(stream)
        t0 = time.perf_counter()
        repeated = 0
        try:
            with _host_blas_threads():
                try:
                    betas, mtx, evs = search.run()
                except _engine.Misprediction:
                    # a kill test decided from a guess that its (device) chain did not confirm: nothing of that search
                    # is kept -- the stream goes back to where the fit began and the search runs again on host chains,
                    # where no decision is taken before its chain has run
                    repeated = 1
                    stream = _capi.LegacyStream(state_at_start)
                    search = new_search(stream, device_chains=False)
                    betas, mtx, evs = search.run()
        finally:
            if rng_state is None:
                stream.publish()       # numpy's global stream ends where the reference's would
            self._rng_state_after = stream.as_numpy_state()
        self.fit_stats = dict(search.stats, seconds=time.perf_counter() - t0, searches_repeated=repeated)
        self.fit_trace = search.trace
        self.fit_substage_stats = search.substage_stats
        _engine._mark('search_end')
        _engine._flush_marks()

        self.betas = betas
        self.avg_betas = np.mean(self.betas, axis=0)
        self.mtx = mtx
        self.evs = evs
        return self.betas, self.mtx, self.evs

    def _check_relats(self, m):
        """``relats_in`` is accepted; only the behaviours the reference actually has are reproduced (FR:1566-1585):
        empty list or a flat list of non-zero ints -> no exclusions; a 2-D list raises like the reference."""
        relats_in = self.relats_in
        if not all(isinstance(v, int) for v in relats_in):
            if np.any(relats_in):
                raise TypeError("relats_in: 2-D exclusion matrices raise TypeError in the reference (FR:1569); "
                                "pass [] to exclude no terms")
            return
        if sum(np.logical_not(relats_in)) != 0:
            raise NameError("relats_in: a flat list containing zeros reaches an undefined 'relats' in the reference "
                            "(FR:1623-1628); pass [] to exclude no terms")

    # -----------------------------------------------------------------------------------------------------
    # evaluate / coverage3
    # -----------------------------------------------------------------------------------------------------

    def evaluate(self, inputs=None, betas=None, mtx=None, draws=None, **kwargs):
        """
        Posterior-mean prediction (and optionally 95 % bounds) at ``inputs`` (FR:851-980).
        Keywords: minmax, draws, clean, ReturnBounds (+ the keywords of ``clean``).
        """
        if not hasattr(self, 'minmax'):
            raise ValueError("To set minmax manually call model.minmax = ([input_min, input_max],[data_min, "
                             "data_max],...) or set clean=True to automtically define min and max from model.inputs")
        default = {'minmax': None, 'draws': self.draws, 'clean': False, 'ReturnBounds': False,
                   '_suppress_normalization_warning': False, 'betas': self.betas, 'mtx': self.mtx}
        default_for_clean = dict(_CLEAN_DEFAULTS)
        default_for_clean['minmax'] = self.minmax
        current = _process_kwargs(_merge_dicts(default, default_for_clean), kwargs)
        for flag in ('clean', 'ReturnBounds'):
            current[flag] = _str_to_bool(current[flag])
        kwargs_to_clean = {kw: current.pop(kw) for kw in list(default_for_clean.keys())}
        if current['draws'] < 40 and current['ReturnBounds']:
            warnings.warn("'draws' must be greater than or equal to 40 to calculate 95% confidence interval bounds.'.")
        if betas is None:
            betas = self.betas
        if draws is None:
            draws = self.draws
        elif betas.shape[0] < draws:
            raise ValueError(f"The number of draws: {draws}  exceeds the number of draws in betas: {betas.shape[0]}"
                             f", \n       draws must be < betas.")
        if mtx is None:
            mtx = self.mtx
        else:
            if isinstance(mtx, int):
                mtx = [mtx]
            mtx = np.array(mtx)
            if mtx.ndim == 1:
                mtx = mtx[np.newaxis, :]
                warnings.warn("Assuming 'mtx' represents a single model. If meant to represent several models, then "
                              "explicitly enter a 2D numpy array where rows correspond to models.")

        if inputs is None:
            if current['clean']:
                warnings.warn("Cleaning was already performed on default 'inputs', so overriding 'clean' to False.",
                              category=UserWarning)
                current['clean'] = False
            normputs = self.inputs
        elif current['clean']:
            normputs = self.clean(inputs, kwargs_from_other=kwargs_to_clean)
        else:
            normputs = np.array(inputs)

        n_draws_avail, mbets = np.shape(betas)
        n = np.shape(normputs)[0]
        mputs = int(np.size(normputs) / n)

        if self.setnos is None:
            setnos = np.random.choice(n_draws_avail, draws, replace=False)     # FR:934, consumes the global stream
            self.setnos = setnos
        else:
            setnos = self.setnos
        if draws == 1:
            setnos = [0]

        normputs = np.asarray(normputs, dtype=np.float64).reshape(n, mputs)
        mtx = np.atleast_2d(np.asarray(mtx))
        backend = self._backend()
        self._upload(backend, normputs, np.zeros(n))
        slots = [_capi.SLOT_ONES]
        if mbets > 1:
            pool = _engine.SlotPool(backend, initial=max(64, mbets + 2))
            term_slots = pool.take(mbets - 1)
            backend.build_terms(np.asarray(mtx[:mbets - 1], dtype=np.int32), term_slots)
            slots = slots + term_slots
        chosen = np.ascontiguousarray(np.asarray(betas)[np.asarray(setnos)[:draws], :], dtype=np.float64)

        if current['ReturnBounds'] == True:  # noqa: E712
            cut = int(np.floor(draws * 0.025) + 1)
            mean, bounds = backend.predict(slots, chosen, cut)
            return mean, bounds
        # the mean over the draws of X beta' (FR:958, 978) is X times the mean draw: one pass over the columns instead
        # of one per draw (N = 1e6, 1000 draws: 19 ms -> 0.1 ms on the device); rounding differs in the last bits only
        return backend.predict(slots, np.mean(chosen, axis=0, keepdims=True))

    def coverage3(self, **kwargs):
        """
        Validation helper (FR:982-1200): evaluate the model on ``inputs`` (default: the stored dataset), optionally
        plot, and return ``(mean, bounds, rmse)`` -- or ``(mean, rmse)`` with ``ReturnBounds=False``.
        """
        try:
            draws = self.draws
        except Exception:
            raise ValueError("self.draws is undefined, specify number of draws to evaluate as kwarg: draws = ")
        current = _process_kwargs({
            'inputs': None, 'data': None, 'draws': self.draws, 'betas': self.betas,
            'plot': False, 'bounds': True, 'xaxis': False, 'labels': True, 'xlabel': 'Index', 'ylabel': 'Data',
            'title': 'FoKL', 'legend': True, 'LegendLabelFoKL': 'FoKL', 'LegendLabelData': 'Data',
            'LegendLabelBounds': 'Bounds', 'ReturnBounds': True,
            'PlotTypeFoKL': 'b', 'PlotSizeFoKL': 2, 'PlotTypeBounds': 'k--', 'PlotSizeBounds': 2,
            'PlotTypeData': 'ro', 'PlotSizeData': 2}, kwargs)
        if isinstance(current['plot'], str):
            if current['plot'].lower() in ['sort', 'sorted', 'order', 'ordered']:
                current['plot'] = 'sorted'
                if current['xlabel'] == 'Index':
                    current['xlabel'] = 'Index (Sorted)'
            else:
                warnings.warn("Keyword input 'plot' is limited to True, False, or 'sorted'.", category=UserWarning)
                current['plot'] = False
        else:
            current['plot'] = _str_to_bool(current['plot'])
        for flag in ('bounds', 'labels', 'legend'):
            current[flag] = _str_to_bool(current[flag])
        if current['labels']:
            for label in ('xlabel', 'ylabel', 'title'):
                if current[label] and not isinstance(current[label], str):
                    current[label] = str(current[label])

        warn_plot = ' and ignoring plot.' if current['plot'] else '.'
        for mine, other in (('inputs', 'data'), ('data', 'inputs')):
            if current[mine] is not None and current[other] is None:
                warnings.warn(f"Keyword argument '{other}' should be defined to align with user-defined '{mine}'. "
                              f"Ignoring RMSE calculation{warn_plot}", category=UserWarning)
                current['data'] = False
        if current['data'] is False and current['plot'] == 'sorted':
            warnings.warn("Keyword argument 'data' must correspond with 'inputs' if requesting a sorted plot. "
                          "Returning a regular plot instead.", category=UserWarning)
            current['plot'] = True

        if current['inputs'] is None:
            current['inputs'] = self.inputs
        if current['data'] is None:
            current['data'] = self.data

        normputs, data, draws = current['inputs'], current['data'], current['draws']
        if draws > np.shape(current['betas'])[0]:
            raise ValueError(f"Number of draws called ({draws}) exceeds number of rows of betas "
                             f"({np.shape(current['betas'])[0]}) ")
        bounds = None
        if current['ReturnBounds'] == True:  # noqa: E712
            mean, bounds = self.evaluate(normputs, betas=current['betas'], draws=draws, ReturnBounds=1,
                                         _suppress_normalization_warning=True)
        else:
            mean = self.evaluate(normputs, betas=current['betas'], draws=draws, ReturnBounds=0,
                                 _suppress_normalization_warning=True)

        if current['plot']:
            self._coverage_plot(current, normputs, data, mean, bounds)

        if data is not False:
            # FR:1193 evaluates sqrt(mean(mean - data) ** 2) over an (n,) - (n, 1) broadcast, i.e. an n x n array
            # whose mean is mean(mean) - mean(data); computed here in O(n).
            rmse = np.sqrt((np.mean(mean) - np.mean(data)) ** 2)
        else:
            rmse = []
        if current['ReturnBounds'] == True:  # noqa: E712
            return mean, bounds, rmse
        return mean, rmse

    def _coverage_plot(self, current, normputs, data, mean, bounds):
        import matplotlib.pyplot as plt
        n = np.shape(normputs)[0]
        xaxis = current['xaxis']
        if xaxis is not False and not isinstance(xaxis, int):
            if len(xaxis) != n:
                warnings.warn("Keyword argument 'xaxis' is limited to an integer indexing the input variable to "
                              "plot along the x-axis (e.g., 0, 1, 2, etc.) or to a vector corresponding to 'data'. "
                              "Leave blank (i.e., False) to plot indices along the x-axis.", category=UserWarning)
                xaxis = False
        if xaxis is False:
            plt_x = np.linspace(0, n - 1, n)
        elif isinstance(xaxis, int):
            try:
                lo, hi = self.minmax[xaxis][0], self.minmax[xaxis][1]
                plt_x = np.array(normputs)[:, xaxis] * (hi - lo) + lo
            except Exception:
                warnings.warn(f"Keyword argument 'xaxis'={xaxis} failed to index 'inputs'. Plotting indices instead.",
                              category=UserWarning)
                plt_x = np.linspace(0, n - 1, n)
        else:
            plt_x = xaxis
        plt_mean, plt_bounds, plt_data = mean, bounds, data
        if current['plot'] == 'sorted':
            order = np.argsort(np.squeeze(data))
            plt_mean, plt_data = mean[order], data[order]
            plt_bounds = bounds[order] if bounds is not None else None
        plt.figure()
        plt.plot(plt_x, plt_mean, current['PlotTypeFoKL'], linewidth=current['PlotSizeFoKL'],
                 label=current['LegendLabelFoKL'])
        if data is not False:
            plt.plot(plt_x, plt_data, current['PlotTypeData'], markersize=current['PlotSizeData'],
                     label=current['LegendLabelData'])
        if plt_bounds is not None and current['bounds']:
            plt.plot(plt_x, plt_bounds[:, 0], current['PlotTypeBounds'], linewidth=current['PlotSizeBounds'],
                     label=current['LegendLabelBounds'])
            plt.plot(plt_x, plt_bounds[:, 1], current['PlotTypeBounds'], linewidth=current['PlotSizeBounds'])
        if current['labels']:
            if current['xlabel']:
                plt.xlabel(current['xlabel'])
            if current['ylabel']:
                plt.ylabel(current['ylabel'])
            if current['title']:
                plt.title(current['title'])
        if current['legend']:
            plt.legend()
        plt.show()

    # -----------------------------------------------------------------------------------------------------
    # housekeeping
    # -----------------------------------------------------------------------------------------------------

    def clear(self, keep=None, clear=None, all=False):
        """Delete every attribute except hyper-parameters and settings (FR:1762-1794)."""
        if all is not False:
            all = _str_to_bool(all)
        if all is False:
            attrs_to_keep = self.keep
            if isinstance(keep, (list, str)):
                attrs_to_keep += keep
                attrs_to_keep = list(np.unique(attrs_to_keep))
            if isinstance(clear, (list, str)):
                for attr in clear:
                    attrs_to_keep.remove(attr)
        else:
            attrs_to_keep = []
        for attr in list(vars(self).keys()):
            if attr not in attrs_to_keep:
                delattr(self, attr)

    def save(self, filename=None, directory=None):
        """Pickle the whole model to a ``.fokl`` file and return its path (FR:1807-1846)."""
        if filename is None:
            filename = 'model_' + time.strftime('%Y%m%d%H%M%S', time.gmtime()) + '.fokl'
        elif filename[-5::] != '.fokl':
            filename = filename + '.fokl'
        path = os.path.join(directory, filename) if directory is not None else filename
        state = copy.copy(self)
        for transient in ('_backend_override', '_comm', '_rng_state_after', '_staged_upload'):
            if hasattr(state, transient):
                delattr(state, transient)
        with open(path, 'wb') as fh:
            pickle.dump(state, fh)
        time.sleep(1)          # so that the next default file name differs (FR:1844)
        return path

    def bss_derivatives(self, **kwargs):
        """
        Gradient (and optionally second partial derivatives) of the fitted model with respect to the inputs
        (FR:594-805).  Keywords: inputs, kernel, d1, d2, draws, betas, phis, mtx, minmax, IndividualDraws,
        ReturnFullArray, ReturnBasis -- same meaning, defaults and output shapes as the reference.

        On the device: for every requested (input m, order) pair the terms that contain x_m are rebuilt with the
        factor of x_m replaced by its derivative (``fokl_build_terms_deriv``), and the columns are contracted with
        the draws (``fokl_predict`` for the mean over draws).
        """
        current = _process_kwargs({'inputs': None, 'kernel': self.kernel, 'd1': None, 'd2': None, 'draws': self.draws,
                                   'betas': None, 'phis': None, 'mtx': self.mtx, 'minmax': self.minmax,
                                   'IndividualDraws': False, 'ReturnFullArray': False, 'ReturnBasis': False}, kwargs)
        for flag in ('IndividualDraws', 'ReturnFullArray', 'ReturnBasis'):
            current[flag] = _str_to_bool(current[flag])
        inputs = self.inputs if current['inputs'] is None else current['inputs']
        betas = self.betas if current['betas'] is None else current['betas']
        phis = self.phis if current['phis'] is None else current['phis']
        kernel, draws, mtx, span = current['kernel'], current['draws'], current['mtx'], current['minmax']

        inputs = np.array(inputs)
        if inputs.ndim == 1:
            inputs = inputs[:, np.newaxis]
        if isinstance(betas, list):
            betas = np.array(betas)
            if betas.ndim == 1:
                betas = betas[:, np.newaxis]
        if isinstance(mtx, int):
            mtx = np.array(mtx)[np.newaxis, np.newaxis]
        else:
            mtx = np.array(mtx)
            if mtx.ndim == 1:
                mtx = mtx[:, np.newaxis]
        if len(span) == 2 and not isinstance(span[0], (list, np.ndarray)):
            span = [span]
        if np.max(np.max(inputs)) > 1 or np.min(np.min(inputs)) < 0:
            warnings.warn("Input 'inputs' should be normalized (0-1). Auto-normalization is in-development.",
                          category=UserWarning)

        N = np.shape(inputs)[0]
        B, M = np.shape(mtx)
        if B != np.shape(betas)[1] - 1:
            betas = np.transpose(betas)
            if B != np.shape(betas)[1] - 1:
                raise ValueError("The shape of 'betas' does not align with the shape of 'mtx'. Transposing did not "
                                 "fix this.")

        derv = []
        for which, di in enumerate((current['d1'], current['d2'])):
            ok = False
            if di is None:
                di, ok = (np.ones(M, dtype=bool) if which == 0 else np.zeros(M, dtype=bool)), True
            elif isinstance(di, str):
                di, ok = (np.ones(M, dtype=bool) if _str_to_bool(di) else np.zeros(M, dtype=bool)), True
            elif isinstance(di, list):
                if len(di) == 1:
                    di = di[0]
                elif len(di) == M:
                    di, ok = np.array(di) != 0, True
                else:
                    raise ValueError("Keyword input 'd1' and/or 'd2', if entered as a list, must be of equal length to "
                                     "the number of input variables.")
            if isinstance(di, bool):
                di, ok = np.ones(M, dtype=bool) * di, True
            elif isinstance(di, int):
                which_input, di = di, np.zeros(M, dtype=bool)
                di[which_input] = True
                ok = True
            if not ok:
                raise ValueError("Keyword input 'd1' and/or 'd2' is limited to an integer indexing an input variable, "
                                 "or to a list of booleans corresponding to the input variables.")
            derv.append(di)
        orders = [di for di in (0, 1) if any(derv[di])]
        if not orders:
            warnings.warn("Function 'bss_derivatives' was called but no derivatives were requested.",
                          category=UserWarning)
            return

        span_m = [span[m][1] - span[m][0] for m in range(M)]
        kid = getKernels.KERNEL_SPLINES if kernel == self.kernels[0] else getKernels.KERNEL_BERNOULLI
        L_phis = len(phis[0][0]) if kid == getKernels.KERNEL_SPLINES else 1
        inputs64 = np.ascontiguousarray(inputs, dtype=np.float64)
        if kid == getKernels.KERNEL_SPLINES:
            self._inputs_to_phind(inputs64, phis, kernel)                # range validation (FR:590-591)
        backend = self._backend()
        packed, nb, width = getKernels.pack_phis(phis, kid)
        backend.upload(inputs64, np.zeros(N), kid, packed, nb, width)
        pool = _engine.SlotPool(backend, initial=max(64, B + 2))

        individual = current['IndividualDraws'] or not draws > 1
        # mean over the draws and the compact output: the columns are kept as they come and assembled once (the
        # general [N, M, 2, draws] array costs more host time than all the device work)
        compact = not individual and not current['ReturnFullArray']
        columns = {}
        dy = None if compact else np.zeros([N, M, 2, draws if individual else 1])
        coef_all = np.asarray(betas)[-draws:, :]
        for m in range(M):
            rows = [b for b in range(B) if int(mtx[b, m]) != 0]         # other terms do not depend on x_m (FR:785-787)
            if not rows:
                continue
            for di in orders:
                if not derv[di][m]:
                    continue
                span_L = span_m[m] / L_phis
                divisor = [1, span_L, span_L ** 2][di + 1]                # FR:758-759
                slots = pool.take(len(rows))
                backend.build_terms_deriv(np.asarray(mtx[rows], dtype=np.int32), slots, m, di + 1, divisor)
                coef = np.ascontiguousarray(coef_all[:, [b + 1 for b in rows]], dtype=np.float64)
                if individual:
                    cols = np.stack([backend.read_slot(s) for s in slots], axis=1)
                    dy[:, m, di, :] = cols @ coef.T
                else:
                    # mean over the draws (FR:793-794) = the columns times the mean draw
                    vec = backend.predict(slots, np.mean(coef, axis=0, keepdims=True))
                    if compact:
                        columns[(di, m)] = vec
                    else:
                        dy[:, m, di, 0] = vec
                pool.give(slots)

        if compact:
            # first derivatives of inputs 0 .. M-1, then second derivatives, all-zero columns dropped (FR:797-800)
            kept = [columns[key] for key in sorted(columns) if np.any(columns[key] != 0)]
            dy = np.stack(kept, axis=1) if kept else np.zeros((N, 0))
        elif not current['ReturnFullArray']:
            dy = np.concatenate([dy[:, :, 0, :], dy[:, :, 1, :]], axis=1)
            dy = dy[:, ~np.all(dy == 0, axis=0)]
        dy = np.squeeze(dy)

        if current['ReturnBasis']:
            # the reference keeps overwriting `basis[n]` inside its loops (FR:783-784); what survives is the basis
            # function of the last (term, input) pair it visits
            basis = np.zeros(N)
            last_m = max(m for m in range(M) if any(derv[di][m] for di in orders))
            b = B - 1
            visited = []
            for md in range(M):
                if int(mtx[b, md]):
                    visited.append(md)
                elif md == last_m:
                    break
            if visited:
                md = visited[-1]
                num = int(mtx[b, md]) - 1
                if kid == getKernels.KERNEL_SPLINES:
                    X, phind, _ = self._inputs_to_phind(inputs64, phis, kernel)
                    for n in range(N):
                        c = [phis[num][k][int(phind[n, md])] for k in range(4)]
                        basis[n] = self.evaluate_basis(c, X[n, md], kernel=kernel)
                else:
                    for n in range(N):
                        basis[n] = self.evaluate_basis(phis[num], inputs64[n, md], kernel=kernel)
            return dy, basis
        return dy

    def fitupdate(self, inputs, data):
        """Sequential-updating fit on already cleaned ``inputs`` / ``data`` (FR:1850-2583; ``fit`` calls it when
        ``update=True``, FR:1365-1367): 2-way sub-stages (ind - i, i) without kill tests, every model scored by the best
        log-likelihood among its draws.  A model without a prior (``built`` False) takes gibbs_Xin_update "case 1"
        (device K1 / K2 / K3 + the native eigenbasis chain); a built model takes its priors from ``betas[burn:-1]`` of
        the previous fit and cases 2 / 3 (device K1 / K2, host samplers on the Gram).  Returns ``(betas, mtx, evs)``
        with ALL burnin + draws rows of the best model's draws, as the reference does; sets ``built`` when the first
        search ends by the tolerance rule (FR:2565).  ``relats_in`` excludes nothing in any variant the reference can run (FR:2449-2466,
        2505-2510: arrays with one row or lists of non-zero ints leave `mrel` at 0; the others raise TypeError)."""
        self.inputs, self.data = inputs, data
        backend = self._backend()
        self._upload(backend, inputs, data)
        return self._update_search(backend, np.shape(inputs)[0], np.shape(inputs)[1])

    def _update_search(self, backend, n, m):
        relats_in = self.relats_in
        if not all(isinstance(v, int) for v in relats_in):                # FR:2449-2466
            if np.all(np.sum(np.logical_not(relats_in), axis=0)):
                raise TypeError("only integer scalar arrays can be converted to a scalar index")
        elif sum(np.logical_not(relats_in)) != 0:                          # FR:2468, 2505-2507: relats_in[t, :] on a list
            raise TypeError("list indices must be integers or slices, not tuple")
        t0 = time.perf_counter()
        if self.built:
            # priors from the previous posterior (gibbs_Xin_update cases 2 / 3): host samplers on device-built Grams,
            # drawing from numpy's global generator call for call as the reference does
            with _host_blas_threads():
                betas, mtx, evs, stats, trace = _update.fit_update_next(
                    backend, n, m, len(self.phis), self.betas, self.burn, self.a, self.b, self.atau, self.btau,
                    self.tolerance, self.burnin + self.draws, self.gimmie, self.aic, self.sigsqd0,
                    console=self.ConsoleOutput)
            self.fit_stats = dict(stats, seconds=time.perf_counter() - t0)
            self.fit_trace = trace
            self.avg_betas = np.mean(betas, axis=0)
            return betas, mtx, evs
        stream = _capi.LegacyStream()
        try:
            with _host_blas_threads():
                betas, mtx, evs, built, stats, trace = _update.fit_update_first(
                    backend, n, m, len(self.phis), self.a, self.b, self.atau, self.btau, self.tolerance,
                    self.burnin + self.draws, self.gimmie, self.aic, self.sigsqd0, stream,
                    console=self.ConsoleOutput)
        finally:
            stream.publish()
        self.built = bool(built)
        self.fit_stats = dict(stats, seconds=time.perf_counter() - t0)
        self.fit_trace = trace
        self.avg_betas = np.mean(betas, axis=0)
        return betas, mtx, evs

    def to_pyomo(self, *args, **kwargs):
        raise NotImplementedError("to_pyomo (FR:1796-1805) is outside the scope of this build")
