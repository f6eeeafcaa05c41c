"""torch-CPU float32 variant of the oracle, for the CPU baseline of bench.py only.

TEST / MEASUREMENT INFRASTRUCTURE -- never imported by the product (mbexwn_vocoder_amd/); only ``bench.py``'s
``cpu_baseline`` leg and ``tests/`` may use it ("the build's own CPU restatement in torch-CPU float32",
because TensorFlow cannot run here or on the GPU box).

``TorchOracleModel`` is ``OracleModel`` (oracle/mbexwn_oracle.py, the numpy restatement of the reference graph) with the
WaveNet -- 98 % of the path's FLOPs (reference MBExWN_NVoc/vocoder/model/custom_AE_layers.py:273-346) -- evaluated with
torch CPU ops in float32: the dilated convolutions as three shifted ``matmul`` calls (MKL, all threads), the gate with
``torch.tanh`` / ``torch.sigmoid`` (vectorised and threaded, where numpy's element-wise functions run on one core).
Everything else (sub-nets, oscillator, PQMF, STFT filter: 2 % of the FLOPs) stays the float32 numpy port.  Same graph, same
weights; a CPU test holds it to the float64 oracle at the path's tolerance.
"""
import numpy as np

from .mbexwn_oracle import OracleModel


class TorchOracleModel(OracleModel):
    def __init__(self, config, raw_weights, wavetables):
        super().__init__(config, raw_weights, wavetables, dtype=np.float32)
        import torch
        self._torch = torch
        self._tw = {}

    def _tweight(self, name):
        if name not in self._tw:
            w, b = self.weight(name)
            self._tw[name] = (self._torch.from_numpy(np.ascontiguousarray(w, dtype=np.float32)),
                              self._torch.from_numpy(np.ascontiguousarray(b, dtype=np.float32)))
        return self._tw[name]

    def wavenet(self, x, mel, return_layers=False, prefix="wn.", channels=None, rate_factor=1):
        """custom_AE_layers.py:273-346 for the canonical options (one channel group, SAME padding, gtu / gfu / gsu / glu gate);
        anything else -- and the stage outputs the parity tests ask for -- goes to the numpy implementation."""
        torch = self._torch
        wn = self.wn
        if (return_layers or int(wn.get("n_ch_groups", 1)) != 1 or str(wn.get("padding", "SAME")).upper() != "SAME" or
                wn.get("disable_conditioning", False)):
            return super().wavenet(x, mel, return_layers=return_layers, prefix=prefix, channels=channels,
                                   rate_factor=rate_factor)
        C = wn["n_channels"] if channels is None else channels
        L = wn.get("n_layers", 12)
        act = wn.get("activation", "gtu")
        with torch.no_grad():
            xt = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32))
            ws, bs = self._tweight(prefix + "start")
            h = xt @ ws[0] + bs                                                    # :280
            cond = torch.from_numpy(np.ascontiguousarray(self.conditioning(mel, prefix, rate_factor), dtype=np.float32))
            out = None
            T = h.shape[1]
            for ll in range(L):
                w, b = self._tweight(f"{prefix}conv1D_{ll}")
                d = self.dilation(ll)
                hp = torch.nn.functional.pad(h, (0, 0, d, d))                      # zero "same" padding, k = 3
                z = hp[:, 0:T] @ w[0] + hp[:, d:d + T] @ w[1] + hp[:, 2 * d:2 * d + T] @ w[2] + b + cond   # :307-309
                zt = z[..., :C]
                if act == "gtu":
                    half = torch.tanh(zt)
                elif act == "gfu":
                    half = zt / (1 + zt.abs())
                elif act == "gsu":
                    half = zt / (1 + zt.abs().sqrt())
                else:                                                              # glu: linear half
                    half = zt
                a = half * torch.sigmoid(z[..., C:])                               # :320-321
                w, b = self._tweight(f"{prefix}res_skip_{ll}")
                r = a @ w[0] + b                                                   # :324
                if ll < L - 1:
                    h = h + r[..., :C]                                             # :326-328
                    s = r[..., C:]
                else:
                    s = r                                                          # :330
                out = s if out is None else out + s                                # :332-335
            w, b = self._tweight(prefix + "end")
            return (out @ w[0] + b).numpy()                                        # :337-340
