"""A numpy stand-in for the part of the TensorFlow/Keras API that the reference's inference
graph touches -- GOLDEN-VECTOR GENERATION ONLY (never imported by the product or by tests).

The reference (roebel/MBExWN_Vocoder) is pure Python on top of TensorFlow, and TensorFlow cannot
be installed in the build container.  To still pin the oracle against *the
reference's own model code*, ``make_reference_forward.py`` imports the reference package from
/root/reference with this module registered as ``tensorflow`` and runs ``MBExWN.call`` unmodified.
Every op below restates the published TensorFlow semantics of the op of the same name
(NWC convolutions are cross-correlations, "SAME" puts the odd padding sample on the right,
``tf.signal`` windows are periodic Hann, ``tf.cumsum`` is a running sum, ``tf.round`` rounds half
to even, ``tf.gather(axis=0, batch_dims=k)`` ignores batch_dims, ...).

FLOAT selects what ``tf.float32`` means: np.float32 (emulation of the float32 TF-CPU run) or
np.float64 (structural run: same graph, rounding noise removed).

Nothing in here is derived from TensorFlow source code; it is written against the public API
documentation.  It is test infrastructure and deliberately minimal.
"""
import abc
import sys
import types

import numpy as np

FLOAT = np.float32


def set_float(dtype):
    global FLOAT
    FLOAT = dtype
    tf.float32 = dtype
    tf.complex64 = np.complex64 if dtype == np.float32 else np.complex128


# ------------------------------------------------------------------------------------------
# tensors
# ------------------------------------------------------------------------------------------
class Shape(tuple):
    """TensorShape look-alike: a tuple that concatenates with lists as well."""

    def __add__(self, other):
        return Shape(tuple(self) + tuple(other))

    def __radd__(self, other):
        return Shape(tuple(other) + tuple(self))

    def __getitem__(self, item):
        res = tuple.__getitem__(self, item)
        return Shape(res) if isinstance(item, slice) else res

    def as_list(self):
        return list(self)


class Tensor(np.ndarray):
    def __new__(cls, value, dtype=None):
        return np.asarray(value, dtype=dtype).view(cls)

    def numpy(self):
        return np.asarray(self)

    @property
    def shape(self):
        return Shape(np.ndarray.shape.__get__(self))

    def assign(self, value):
        np.asarray(self)[...] = np.asarray(value, dtype=self.dtype)
        return self

    def __array_ufunc__(self, ufunc, method, *inputs, **kwargs):
        """TF converts python / numpy scalars that meet a tensor to the tensor's dtype
        (no silent float64 promotion): do the same before handing over to numpy."""
        real = None
        for ii in inputs:
            if isinstance(ii, Tensor) and ii.dtype.kind in "fc":
                real = np.float32 if ii.dtype in (np.float32, np.complex64) else np.float64
                break
        args = []
        for ii in inputs:
            if isinstance(ii, Tensor):
                args.append(np.asarray(ii))
            elif real is not None and isinstance(ii, np.generic) and ii.dtype.kind == "f":
                args.append(real(ii))
            else:
                args.append(ii)
        if "out" in kwargs:
            kwargs["out"] = tuple(np.asarray(oo) if isinstance(oo, Tensor) else oo for oo in kwargs["out"])
        res = getattr(ufunc, method)(*args, **kwargs)
        if isinstance(res, tuple):
            return tuple(_t(rr) if isinstance(rr, (np.ndarray, np.generic)) else rr for rr in res)
        if isinstance(res, (np.ndarray, np.generic)):
            return _t(res)
        return res


def _t(x, dtype=None):
    return Tensor(x, dtype)


def _dt(dtype):
    if dtype is None:
        return None
    if dtype is float:
        return FLOAT
    return dtype


# ------------------------------------------------------------------------------------------
# module tree
# ------------------------------------------------------------------------------------------
tf = types.ModuleType("tensorflow")
tf.__path__ = []
tf.float32 = FLOAT
tf.float64 = np.float64
tf.float16 = np.float16
tf.int32 = np.int32
tf.int64 = np.int64
tf.bool = np.bool_
tf.complex64 = np.complex64
tf.complex128 = np.complex128
tf.newaxis = None
tf.Tensor = Tensor
tf.DType = type(np.dtype("float32"))


def _constant(value, dtype=None, shape=None, name=None):
    arr = np.asarray(value, dtype=_dt(dtype))
    if dtype is None and arr.dtype == np.float64 and not isinstance(value, (np.ndarray, np.generic)):
        arr = arr.astype(FLOAT)            # python floats become float32 tensors in TF
    if shape is not None:
        arr = np.broadcast_to(arr, shape).copy()
    return _t(arr)


tf.constant = _constant
tf.convert_to_tensor = lambda value, dtype=None, **kw: _constant(value, dtype)


class Variable(Tensor):
    def __new__(cls, initial_value, trainable=None, dtype=None, name=None):
        return np.array(initial_value, dtype=_dt(dtype)).view(cls)


tf.Variable = Variable
tf.cast = lambda x, dtype, name=None: _t(np.asarray(x).astype(_dt(dtype)))
tf.ones = lambda shape, dtype=None, name=None: _t(np.ones(tuple(shape) if not np.isscalar(shape) else (shape,), dtype=_dt(dtype) or FLOAT))
tf.zeros = lambda shape, dtype=None, name=None: _t(np.zeros(tuple(shape) if not np.isscalar(shape) else (shape,), dtype=_dt(dtype) or FLOAT))
tf.range = lambda *a, dtype=None, **kw: _t(np.arange(*a, dtype=_dt(dtype)))
tf.shape = lambda x: Shape(np.asarray(x).shape)
tf.concat = lambda values, axis, name=None: _t(np.concatenate([np.asarray(v) for v in values], axis=axis))
# tf.tensordot(a, b, axes): contraction of the last `axes` axes of a with the first `axes` of b (numpy's definition is the
# published one); a float constant b takes the float type of a, like TensorFlow's implicit conversion of numpy arguments
tf.tensordot = lambda a, b, axes: _t(np.tensordot(np.asarray(a), np.asarray(b).astype(np.asarray(a).dtype), axes=axes))
tf.stack = lambda values, axis=0, name=None: _t(np.stack([np.asarray(v) for v in values], axis=axis))
tf.tile = lambda x, multiples, name=None: _t(np.tile(np.asarray(x), tuple(multiples)))
tf.reshape = lambda x, shape, name=None: _t(np.reshape(np.asarray(x), tuple(shape)))
tf.transpose = lambda x, perm=None: _t(np.transpose(np.asarray(x), perm))
tf.expand_dims = lambda x, axis: _t(np.expand_dims(np.asarray(x), axis))
tf.stop_gradient = lambda x: x
tf.repeat = lambda x, repeats, axis=None: _t(np.repeat(np.asarray(x), repeats, axis=axis))
tf.abs = lambda x: _t(np.abs(np.asarray(x)))
tf.exp = lambda x: _t(np.exp(np.asarray(x)))
tf.sqrt = lambda x: _t(np.sqrt(np.asarray(x)))
tf.square = lambda x: _t(np.square(np.asarray(x)))
tf.floor = lambda x: _t(np.floor(np.asarray(x)))
tf.round = lambda x: _t(np.rint(np.asarray(x)))            # round half to even, like TF
tf.sin = lambda x: _t(np.sin(np.asarray(x)))
tf.cos = lambda x: _t(np.cos(np.asarray(x)))
tf.pow = lambda x, y: _t(np.power(np.asarray(x), y))
tf.maximum = lambda x, y: _t(np.maximum(np.asarray(x), np.asarray(y, dtype=np.asarray(x).dtype) if np.isscalar(y) else np.asarray(y)))
tf.minimum = lambda x, y: _t(np.minimum(np.asarray(x), np.asarray(y, dtype=np.asarray(x).dtype) if np.isscalar(y) else np.asarray(y)))
tf.reduce_sum = lambda x, axis=None, keepdims=False: _t(np.sum(np.asarray(x), axis=axis if axis is None or np.isscalar(axis) else tuple(axis), keepdims=keepdims))
tf.reduce_mean = lambda x, axis=None, keepdims=False: _t(np.mean(np.asarray(x), axis=axis if axis is None or np.isscalar(axis) else tuple(axis), keepdims=keepdims))
tf.reduce_max = lambda x, axis=None, keepdims=False: _t(np.max(np.asarray(x), axis=axis, keepdims=keepdims))
tf.reduce_min = lambda x, axis=None, keepdims=False: _t(np.min(np.asarray(x), axis=axis, keepdims=keepdims))
tf.reduce_all = lambda x, axis=None: bool(np.all(np.asarray(x)))
tf.complex = lambda re, im: _t(np.asarray(re) + 1j * np.asarray(im)).astype(tf.complex64)
tf.print = print


def _split(value, num_or_size_splits, axis=0, num=None, name=None):
    value = np.asarray(value)
    if np.isscalar(num_or_size_splits):
        return [_t(v) for v in np.split(value, int(num_or_size_splits), axis=axis)]
    idx = np.cumsum(num_or_size_splits)[:-1]
    return [_t(v) for v in np.split(value, idx, axis=axis)]


tf.split = _split


def _pad(tensor, paddings, mode="CONSTANT", constant_values=0, name=None):
    mode = {"CONSTANT": "constant", "REFLECT": "reflect", "SYMMETRIC": "symmetric"}[mode.upper()]
    paddings = [tuple(int(pp) for pp in pair) for pair in paddings]
    kw = {"constant_values": constant_values} if mode == "constant" else {}
    return _t(np.pad(np.asarray(tensor), paddings, mode=mode, **kw))


tf.pad = _pad


def _cumsum(x, axis=0, exclusive=False, reverse=False, name=None):
    x = np.asarray(x)
    res = np.cumsum(x, axis=axis, dtype=x.dtype)       # running (sequential) sum in the tensor's dtype
    if exclusive:
        res = res - x
    return _t(res)


tf.cumsum = _cumsum


def _gather(params, indices, validate_indices=None, axis=None, batch_dims=0, name=None):
    params = np.asarray(params)
    indices = np.asarray(indices)
    if axis is None:
        axis = batch_dims
    if axis == 0:
        # documented quirk of tf.gather's python wrapper: with axis == 0 the op is issued without
        # batch_dims, i.e. a plain take along the first axis.
        return _t(np.take(params, indices, axis=0))
    if batch_dims != 0:
        raise NotImplementedError("gather with batch_dims and axis != 0")
    return _t(np.take(params, indices, axis=axis))


tf.gather = _gather


def _assert_equal(x, y, message=None, **kw):
    if not np.all(np.asarray(x) == np.asarray(y)):
        raise AssertionError(message)


tf.assert_equal = _assert_equal


def _function(func=None, **kwargs):
    if func is not None and callable(func):
        return func
    return lambda ff: ff


tf.function = _function
tf.TensorSpec = lambda *a, **kw: None

# ---- tf.math ----------------------------------------------------------------------------
tf.math = types.ModuleType("tensorflow.math")
tf.math.log = lambda x: _t(np.log(np.asarray(x, dtype=FLOAT) if np.isscalar(x) else np.asarray(x)))
tf.math.tanh = lambda x: _t(np.tanh(np.asarray(x)))
tf.math.real = lambda x: _t(np.real(np.asarray(x)))
tf.math.imag = lambda x: _t(np.imag(np.asarray(x)))
tf.math.is_finite = lambda x: _t(np.isfinite(np.asarray(x)))
tf.math.pow = tf.pow

# ---- tf.linalg --------------------------------------------------------------------------
tf.linalg = types.ModuleType("tensorflow.linalg")
tf.linalg.norm = lambda x, axis=None: _t(np.sqrt(np.sum(np.square(np.asarray(x)), axis=axis)))
tf.linalg.matmul = lambda a, b, transpose_b=False: _t(np.matmul(np.asarray(a), np.swapaxes(np.asarray(b), -1, -2) if transpose_b else np.asarray(b)))

# ---- tf.random --------------------------------------------------------------------------
tf.random = types.ModuleType("tensorflow.random")
INJECTED_NOISE = {"normal": None}


def _random_normal(shape, mean=0.0, stddev=1.0, dtype=None, seed=None, name=None):
    """TensorFlow's Philox stream is not reproducible outside TF: the generator script injects
    the N(0,1) draw that the graph consumes."""
    noise = INJECTED_NOISE["normal"]
    if noise is None:
        raise RuntimeError("no injected noise available")
    noise = np.asarray(noise).reshape(tuple(shape))
    return _t(noise.astype(FLOAT) * stddev + mean)


tf.random.normal = _random_normal
tf.random.uniform = lambda *a, **kw: (_ for _ in ()).throw(NotImplementedError("training only"))
tf.random.set_seed = lambda seed: None


# ---- tf.nn ------------------------------------------------------------------------------
tf.nn = types.ModuleType("tensorflow.nn")
tf.nn.tanh = lambda x: _t(np.tanh(np.asarray(x)))
tf.nn.sigmoid = lambda x: _t(1 / (1 + np.exp(-np.asarray(x))))


def _l2_normalize(x, axis=None, epsilon=1e-12, name=None):
    x = np.asarray(x)
    sq = np.sum(np.square(x), axis=tuple(axis) if not np.isscalar(axis) else axis, keepdims=True)
    return _t(x * (1 / np.sqrt(np.maximum(sq, np.asarray(epsilon, dtype=x.dtype)))))


tf.nn.l2_normalize = _l2_normalize


def _same_pads(length, ksize_eff, stride):
    out = -(-length // stride)
    total = max((out - 1) * stride + ksize_eff - length, 0)
    return total // 2, total - total // 2


def conv1d_nwc(x, w, stride=1, padding="VALID", dilation=1):
    """x (B,T,Cin), w (K,Cin,Cout): y[b,t,co] = sum_{j,ci} x[b, t*stride + j*dilation, ci] w[j,ci,co]."""
    x = np.asarray(x)
    w = np.asarray(w)
    K = w.shape[0]
    keff = (K - 1) * dilation + 1
    padding = padding.upper()
    if padding == "SAME":
        pl, pr = _same_pads(x.shape[1], keff, stride)
        x = np.pad(x, ((0, 0), (pl, pr), (0, 0)))
    elif padding == "CAUSAL":
        x = np.pad(x, ((0, 0), (keff - 1, 0), (0, 0)))
    elif padding != "VALID":
        raise ValueError(padding)
    t_out = (x.shape[1] - keff) // stride + 1
    y = np.zeros((x.shape[0], t_out, w.shape[2]), dtype=np.result_type(x.dtype, w.dtype))
    for jj in range(K):
        seg = x[:, jj * dilation: jj * dilation + (t_out - 1) * stride + 1: stride, :]
        y += seg @ w[jj]
    return y


def _nn_conv1d(input, filters, stride=1, padding="VALID", data_format="NWC", dilations=None, name=None):
    if isinstance(stride, (list, tuple)):
        stride = stride[1]
    if isinstance(dilations, (list, tuple)):
        dilations = dilations[1]
    return _t(conv1d_nwc(input, filters, stride=int(stride), padding=padding, dilation=int(dilations or 1)))


tf.nn.conv1d = _nn_conv1d


def _depthwise_conv2d(input, filter, strides, padding, data_format="NHWC", dilations=None, name=None):
    x = np.asarray(input)                 # (B,H,W,C)
    f = np.asarray(filter)                # (kh,kw,C,mult)
    kh, kw, C, mult = f.shape
    if kh != 1 or list(strides) != [1, 1, 1, 1] or padding.upper() != "SAME":
        raise NotImplementedError
    pl, pr = _same_pads(x.shape[2], kw, 1)
    xp = np.pad(x, ((0, 0), (0, 0), (pl, pr), (0, 0)))
    W = x.shape[2]
    out = np.zeros(x.shape[:3] + (C, mult), dtype=np.result_type(x.dtype, f.dtype))
    for jj in range(kw):
        out += xp[:, :, jj:jj + W, :, None] * f[0, jj][None, None, None, :, :]
    return _t(out.reshape(x.shape[:3] + (C * mult,)))        # channel order c*mult + m


tf.nn.depthwise_conv2d = _depthwise_conv2d


def _conv1d_transpose(input, filters, output_shape, strides, padding="SAME", data_format="NWC", dilations=None,
                      name=None):
    x = np.asarray(input)                 # (B,S,Cin)
    f = np.asarray(filters)               # (K, Cout, Cin)
    K, cout, cin = f.shape
    stride = int(strides if np.isscalar(strides) else strides[1] if len(strides) == 3 else strides[0])
    B, S, _ = x.shape
    full = np.zeros((B, (S - 1) * stride + K, cout), dtype=np.result_type(x.dtype, f.dtype))
    for jj in range(K):
        full[:, jj: jj + (S - 1) * stride + 1: stride, :] += x @ f[jj].T
    out_len = int(output_shape[1])
    if padding.upper() == "SAME":
        total = max(K - stride, 0)
        begin = total // 2
    else:
        begin = 0
    res = full[:, begin: begin + out_len, :]
    if res.shape[1] < out_len:
        res = np.pad(res, ((0, 0), (0, out_len - res.shape[1]), (0, 0)))
    return _t(res)


tf.nn.conv1d_transpose = _conv1d_transpose

# ---- tf.signal --------------------------------------------------------------------------
tf.signal = types.ModuleType("tensorflow.signal")


def _hann_window(window_length, periodic=True, dtype=None, name=None):
    dtype = _dt(dtype) or FLOAT
    even = 1 - window_length % 2
    n = dtype(window_length + int(periodic) * even - 1)
    count = np.arange(window_length, dtype=dtype)
    arg = dtype(2 * np.pi) * count / n
    return _t((dtype(0.5) - dtype(0.5) * np.cos(arg)).astype(dtype))


tf.signal.hann_window = _hann_window


def _frame(signal, frame_length, frame_step):
    signal = np.asarray(signal)
    n = 1 + (signal.shape[-1] - frame_length) // frame_step
    idx = np.arange(frame_length)[None, :] + frame_step * np.arange(n)[:, None]
    return signal[..., idx]


def _rfft(x, fft_length=None, name=None):
    x = np.asarray(x)
    n = x.shape[-1] if fft_length is None else int(np.asarray(fft_length).reshape(-1)[0])
    cdt = np.complex64 if x.dtype == np.float32 else np.complex128
    return _t(np.fft.rfft(x, n=n, axis=-1).astype(cdt))


tf.signal.rfft = _rfft
tf.signal.fft = lambda x, name=None: _t(np.fft.fft(np.asarray(x), axis=-1))


def _stft(signals, frame_length, frame_step, fft_length=None, window_fn=_hann_window, pad_end=False, name=None):
    if pad_end:
        raise NotImplementedError
    signals = np.asarray(signals)
    frames = _frame(signals, frame_length, frame_step)
    if window_fn is not None:
        frames = frames * np.asarray(window_fn(frame_length, dtype=signals.dtype.type))
    return _rfft(frames.astype(signals.dtype), fft_length=fft_length)


tf.signal.stft = _stft


def _overlap_and_add(signal, frame_step, name=None):
    signal = np.asarray(signal)
    n_frames, frame_length = signal.shape[-2:]
    out = np.zeros(signal.shape[:-2] + ((n_frames - 1) * frame_step + frame_length,), dtype=signal.dtype)
    for tt in range(n_frames):
        out[..., tt * frame_step: tt * frame_step + frame_length] += signal[..., tt, :]
    return _t(out)


tf.signal.overlap_and_add = _overlap_and_add


def _inverse_stft_window_fn(frame_step, forward_window_fn=_hann_window, name=None):
    def inverse_window(frame_length, dtype=None):
        dtype = _dt(dtype) or FLOAT
        fwd = np.asarray(forward_window_fn(frame_length, dtype=dtype))
        den = np.square(fwd)
        overlaps = -(-frame_length // frame_step)
        den = np.pad(den, (0, overlaps * frame_step - frame_length))
        den = den.reshape(overlaps, frame_step).sum(axis=0, keepdims=True)
        den = np.tile(den, (overlaps, 1)).reshape(overlaps * frame_step)
        return _t((fwd / den[:frame_length]).astype(dtype))
    return inverse_window


tf.signal.inverse_stft_window_fn = _inverse_stft_window_fn


def _inverse_stft(stfts, frame_length, frame_step, fft_length=None, window_fn=_hann_window, name=None):
    stfts = np.asarray(stfts)
    rdt = np.float32 if stfts.dtype == np.complex64 else np.float64
    n = fft_length if fft_length is not None else 2 * (stfts.shape[-1] - 1)
    frames = np.fft.irfft(stfts, n=n, axis=-1).astype(rdt)[..., :frame_length]
    if frames.shape[-1] < frame_length:
        frames = np.pad(frames, [(0, 0)] * (frames.ndim - 1) + [(0, frame_length - frames.shape[-1])])
    if window_fn is not None:
        frames = frames * np.asarray(window_fn(frame_length, dtype=rdt))
    return _overlap_and_add(frames.astype(rdt), frame_step)


tf.signal.inverse_stft = _inverse_stft

# ---- tf.summary (training only) -----------------------------------------------------------
tf.summary = types.ModuleType("tensorflow.summary")
tf.summary.scalar = lambda *a, **kw: None
tf.summary.histogram = lambda *a, **kw: None
tf.summary.experimental = types.SimpleNamespace(get_step=lambda: 0)

# ------------------------------------------------------------------------------------------
# keras
# ------------------------------------------------------------------------------------------
keras = types.ModuleType("tensorflow.keras")
keras.__path__ = []
layers = types.ModuleType("tensorflow.keras.layers")
activations = types.ModuleType("tensorflow.keras.activations")
initializers = types.ModuleType("tensorflow.keras.initializers")
backend = types.ModuleType("tensorflow.keras.backend")
backend.epsilon = lambda: 1e-7

_name_counts = {}


def _unique(name):
    cnt = _name_counts.get(name, 0)
    _name_counts[name] = cnt + 1
    return name if cnt == 0 else f"{name}_{cnt}"


class Layer(metaclass=abc.ABCMeta):
    def __init__(self, name=None, dtype=None, trainable=True, **kwargs):
        self.name = name if name is not None else _unique(type(self).__name__.lower())
        self.dtype = _dt(dtype) if dtype is not None else FLOAT
        if isinstance(self.dtype, str):
            self.dtype = FLOAT if self.dtype == "float32" else np.dtype(self.dtype).type
        self.trainable = trainable
        self.built = False
        self._weights = []

    def add_weight(self, name=None, shape=None, initializer=None, dtype=None, trainable=True, **kwargs):
        shape = (int(shape),) if np.isscalar(shape) else tuple(int(ss) for ss in shape)
        dtype = _dt(dtype) or self.dtype
        if initializer is None:
            value = np.zeros(shape, dtype=dtype)
        elif callable(initializer):
            value = np.asarray(initializer(shape, dtype), dtype=dtype)
        else:
            raise TypeError(initializer)
        var = _t(value.copy())
        self._weights.append(var)
        return var

    def build(self, input_shape):
        self.built = True

    def compute_output_shape(self, input_shape):
        return input_shape

    def __call__(self, inputs, *args, **kwargs):
        if not self.built:
            if isinstance(inputs, (tuple, list)):
                shape = tuple(Shape(np.asarray(ii).shape) for ii in inputs)
            else:
                shape = Shape(np.asarray(inputs).shape)
            self.build(shape)
            self.built = True
        return self.call(inputs, *args, **kwargs)

    def get_config(self):
        return {"name": self.name}

    @property
    def trainable_weights(self):
        return list(self._weights)


layers.Layer = Layer


class Module(object):
    def __init__(self, name=None):
        self.name = name


tf.Module = Module


class _RandomNormal:
    def __init__(self, mean=0.0, stddev=0.05, seed=None):
        self.mean, self.stddev = mean, stddev

    def __call__(self, shape, dtype=None):
        # weights are overwritten by the generator script; the draw only has to be finite
        rng = np.random.default_rng(0)
        return rng.normal(self.mean, self.stddev, size=shape).astype(_dt(dtype) or FLOAT)


class _Constant:
    def __init__(self, value=0.0):
        self.value = value

    def __call__(self, shape, dtype=None):
        return np.full(shape, self.value, dtype=_dt(dtype) or FLOAT)


initializers.RandomNormal = _RandomNormal
initializers.Constant = _Constant
initializers.constant = _Constant
initializers.get = lambda ident: _Constant(0.0) if ident in ("zero", "zeros") else ident


class Conv1D(Layer):
    def __init__(self, filters, kernel_size, strides=1, padding="valid", dilation_rate=1, activation=None,
                 use_bias=True, kernel_initializer=None, name=None, dtype=None, **kwargs):
        super().__init__(name=name if name is not None else _unique("conv1d"), dtype=dtype)
        self.filters = int(filters)
        self.kernel_size = (int(kernel_size),) if np.isscalar(kernel_size) else tuple(kernel_size)
        self.strides = (int(strides),) if np.isscalar(strides) else tuple(strides)
        self.padding = padding.lower()
        self.dilation_rate = (int(dilation_rate),) if np.isscalar(dilation_rate) else tuple(dilation_rate)
        self.use_bias = use_bias
        self.kernel_initializer = kernel_initializer or _RandomNormal(0.0, 0.05)
        self.kernel = None
        self.bias = None
        if activation is not None:
            raise NotImplementedError

    def build(self, input_shape):
        cin = int(input_shape[-1])
        self.kernel = self.add_weight("kernel", (self.kernel_size[0], cin, self.filters), self.kernel_initializer)
        if self.use_bias:
            self.bias = self.add_weight("bias", (self.filters,), _Constant(0.0))
        self.built = True

    def compute_output_shape(self, input_shape):
        length = input_shape[1]
        if length is not None:
            keff = (self.kernel_size[0] - 1) * self.dilation_rate[0] + 1
            if self.padding == "valid":
                length = (length - keff) // self.strides[0] + 1
            else:
                length = -(-length // self.strides[0])
        return Shape((input_shape[0], length, self.filters))

    def call(self, inputs):
        y = conv1d_nwc(inputs, self.kernel, stride=self.strides[0], padding=self.padding,
                       dilation=self.dilation_rate[0])
        if self.use_bias:
            y = y + np.asarray(self.bias)
        return _t(y)

    def get_config(self):
        return {"name": self.name, "filters": self.filters, "kernel_size": self.kernel_size}


layers.Conv1D = Conv1D


class PReLU(Layer):
    def __init__(self, alpha_initializer=None, shared_axes=None, name=None, **kwargs):
        super().__init__(name=name if name is not None else _unique("p_re_lu"))
        self.alpha_initializer = alpha_initializer or _Constant(0.0)
        self.shared_axes = list(shared_axes) if shared_axes is not None else []
        self.alpha = None

    def build(self, input_shape):
        shape = list(input_shape[1:])
        for ax in self.shared_axes:
            shape[ax - 1] = 1
        self.alpha = self.add_weight("alpha", tuple(shape), self.alpha_initializer)
        self.built = True

    def call(self, inputs):
        x = np.asarray(inputs)
        return _t(np.maximum(x, 0) - np.asarray(self.alpha) * np.maximum(-x, 0))


layers.PReLU = PReLU


class LeakyReLU(Layer):
    def __init__(self, alpha=0.3, name=None, **kwargs):
        super().__init__(name=name if name is not None else _unique("leaky_re_lu"))
        self.alpha = alpha

    def call(self, inputs):
        x = np.asarray(inputs)
        return _t(np.where(x > 0, x, x * np.asarray(self.alpha, dtype=x.dtype)))


layers.LeakyReLU = LeakyReLU


class ReLU(Layer):
    def call(self, inputs):
        return _t(np.maximum(np.asarray(inputs), 0))


layers.ReLU = ReLU

activations.get = lambda ident: (lambda x: x) if ident is None else getattr(activations, ident)
activations.serialize = lambda fn: getattr(fn, "__name__", "linear")
activations.tanh = tf.nn.tanh
activations.sigmoid = tf.nn.sigmoid
activations.softsign = lambda x: _t(np.asarray(x) / (1 + np.abs(np.asarray(x))))
activations.elu = lambda x: _t(np.where(np.asarray(x) > 0, x, np.expm1(x)))
activations.selu = lambda x: (_ for _ in ()).throw(NotImplementedError)
activations.exponential = tf.exp


class Model(Layer):
    pass


keras.Model = Model
keras.layers = layers
keras.activations = activations
keras.initializers = initializers
keras.backend = backend
tf.keras = keras


# ------------------------------------------------------------------------------------------
# librosa stub (only imported, never used on the inference path) and numpy/scipy aliases
# ------------------------------------------------------------------------------------------
def _librosa():
    lib = types.ModuleType("librosa")
    lib.__path__ = []

    class ParameterError(Exception):
        pass

    lib.ParameterError = ParameterError

    def _unavailable(*a, **kw):
        raise NotImplementedError("librosa is not available in this container")

    core = types.ModuleType("librosa.core")
    core.__path__ = []
    convert = types.ModuleType("librosa.core.convert")
    convert.mel_frequencies = _unavailable
    convert.hz_to_mel = _unavailable
    convert.mel_to_hz = _unavailable
    core.convert = convert
    filters = types.ModuleType("librosa.filters")
    filters.mel = _unavailable
    feature = types.ModuleType("librosa.feature")
    feature.melspectrogram = lambda *a, **kw: None
    lib.core, lib.filters, lib.feature = core, filters, feature
    return {"librosa": lib, "librosa.core": core, "librosa.core.convert": convert,
            "librosa.filters": filters, "librosa.feature": feature}


class _Caster(dict):
    def __missing__(self, key):
        return lambda x: np.asarray(x).astype(key)


def install(reference_root="/root/reference"):
    """Register the stand-ins and the removed numpy/scipy aliases the reference still uses
    (np.int, np.float, np.cast, scipy.signal.kaiser/hanning -- SURVEY.md F4), then make the
    reference package importable."""
    import scipy.signal as ss
    sys.modules["tensorflow"] = tf
    sys.modules["tensorflow.keras"] = keras
    sys.modules["tensorflow.keras.layers"] = layers
    sys.modules["tensorflow.keras.activations"] = activations
    sys.modules["tensorflow.keras.initializers"] = initializers
    sys.modules["tensorflow.math"] = tf.math
    sys.modules["tensorflow.signal"] = tf.signal
    sys.modules["tensorflow.nn"] = tf.nn
    for kk, vv in _librosa().items():
        sys.modules.setdefault(kk, vv)
    if not hasattr(np, "int"):
        np.int = int
    if not hasattr(np, "float"):
        np.float = float
    if not hasattr(np, "cast"):
        np.cast = _Caster()
    if not hasattr(ss, "kaiser"):
        ss.kaiser = ss.windows.kaiser
    if not hasattr(ss, "hanning"):
        ss.hanning = ss.windows.hann
    if reference_root not in sys.path:
        sys.path.insert(0, reference_root)
