"""Fully connected networks on the GPU and risk-aware stepping (second half).

The reference wraps the task policy's action in a risk check (actions.py:303-340): a Keras risk network rates (risk observation,
action) (safe_motions_base.py:1498-1502,1598-1603; architecture train_risk_network.py:273-282) and, where the rating reaches
``risk_threshold``, the action of the backup policy - an RLlib PPO agent with the model of
model/keras_fcnet_last_layer_activation.py, run without exploration (safe_motions_base.py:631) - is executed instead.  Here both
networks are ``Mlp`` objects: dense layers evaluated by the library's MFMA kernel over the whole batch inside ``step()``.

Weights are supplied as plain arrays (an ``.npz`` with ``w0, b0, w1, b1, ...`` in Keras layout ``[in][out]``).  Nothing in this
package deserialises the reference's checkpoints: those are pickle files and TF SavedModels of a public repository; a maintainer
exports the arrays once, with the tools that wrote them.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import lib as _lib

ACTIVATIONS = {"linear": 0, None: 0, "tanh": 1, "swish": 2, "relu": 3, "sigmoid": 4, "selu": 5}


class Mlp:
    """``y = f_n(... f_1(x W_1 + b_1) ...)`` on the GPU.  weights[l]: float32 [in_l][out_l]; activations[l]: a name of ACTIVATIONS."""

    def __init__(self, weights, biases, activations, device=0):
        import torch
        if not torch.cuda.is_available():
            raise RuntimeError("Mlp needs a GPU: the HIP library has no CPU path")
        self._L = _lib.load()
        self._torch = torch
        self.device = torch.device("cuda", int(device))
        self.weights = [np.ascontiguousarray(w, dtype=np.float32) for w in weights]
        self.biases = [np.ascontiguousarray(b, dtype=np.float32).reshape(-1) for b in biases]
        self.activations = list(activations)
        n = len(self.weights)
        if not (n == len(self.biases) == len(self.activations)) or n == 0:
            raise ValueError("weights, biases and activations must have the same, non-zero length")
        for l, (w, b) in enumerate(zip(self.weights, self.biases)):
            if w.ndim != 2 or b.shape[0] != w.shape[1] or (l and w.shape[0] != self.weights[l - 1].shape[1]):
                raise ValueError("layer %d: weight %s / bias %s do not chain" % (l, w.shape, b.shape))
        self.sizes = [self.weights[0].shape[0]] + [w.shape[1] for w in self.weights]
        self.in_size, self.out_size = self.sizes[0], self.sizes[-1]
        acts = (C.c_int32 * n)(*[ACTIVATIONS[a] for a in self.activations])
        sizes = (C.c_int32 * (n + 1))(*self.sizes)
        wp = (C.c_void_p * n)(*[w.ctypes.data for w in self.weights])
        bp = (C.c_void_p * n)(*[b.ctypes.data for b in self.biases])
        h = C.c_void_p()
        rc = self._L.smr_mlp_create(n, sizes, wp, bp, acts, int(device), C.byref(h))
        _lib.check(self._L, None, rc, "smr_mlp_create")
        self._h = h

    @classmethod
    def from_npz(cls, path, activations, device=0):
        """arrays w0, b0, w1, b1, ... (Keras Dense kernels / biases) of an .npz file; allow_pickle stays off"""
        z = np.load(path, allow_pickle=False)
        n = len(activations)
        return cls([z["w%d" % i] for i in range(n)], [z["b%d" % i] for i in range(n)], activations, device=device)

    def __call__(self, x):
        """x: float32 tensor [rows, >= in_size] on the network's device (row stride = its second dimension) -> [rows, out_size]"""
        t = self._torch
        x = x.to(device=self.device, dtype=t.float32).contiguous()
        if x.dim() != 2 or x.shape[1] < self.in_size:
            raise ValueError("input must be [rows, >= %d]" % self.in_size)
        y = t.empty((x.shape[0], self.out_size), dtype=t.float32, device=self.device)
        with t.cuda.device(self.device):
            stream = t.cuda.current_stream(self.device).cuda_stream
            rc = self._L.smr_mlp_forward(self._h, C.c_void_p(x.data_ptr()), x.shape[1], C.c_void_p(y.data_ptr()), self.out_size, x.shape[0],
                                         C.c_void_p(stream))
        _lib.check(self._L, None, rc, "smr_mlp_forward")
        return y

    def reference(self, x):
        """the same network in numpy float32 (tests: the plain fp32 reference of the kernel)"""
        h = np.asarray(x, dtype=np.float32)[:, :self.in_size]
        f = {0: lambda v: v, 1: np.tanh, 2: lambda v: v / (1 + np.exp(-v)), 3: lambda v: np.maximum(v, 0),
             4: lambda v: 1 / (1 + np.exp(-v)),
             5: lambda v: np.where(v > 0, 1.05070098735548 * v, 1.05070098735548 * 1.67326324235438 * (np.exp(v) - 1))}
        for w, b, a in zip(self.weights, self.biases, self.activations):
            h = f[ACTIVATIONS[a]]((h @ w + b).astype(np.float32)).astype(np.float32)
        return h

    def close(self):
        if getattr(self, "_h", None):
            self._L.smr_mlp_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def backup_policy(weights, biases, device=0, hidden_activation="swish", last_layer_activation="tanh"):
    """The policy branch of FullyConnectedNetworkLastLayerActivation (model/keras_fcnet_last_layer_activation.py:74-131 with the
    shipped custom_model_config: fcnet_hiddens [256, 128], fcnet_activation swish, last_layer_activation tanh): fc_1, fc_2, fc_out.
    Its first half of outputs are the action means = the deterministic action (explore=False, safe_motions_base.py:631)."""
    acts = [hidden_activation] * (len(weights) - 1) + [last_layer_activation]
    return Mlp(weights, biases, acts, device=device)


def risk_network(weights, biases, device=0, hidden_activation="swish", last_layer_activation="sigmoid"):
    """train_risk_network.py:273-282: Dense(fcnet_hiddens[i], hidden_layer_activation) ... Dense(1, last_layer_activation)"""
    acts = [hidden_activation] * (len(weights) - 1) + [last_layer_activation]
    return Mlp(weights, biases, acts, device=device)
