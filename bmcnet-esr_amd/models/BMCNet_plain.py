"""BMCNet_plain on MI355X: reference-compatible signatures and state_dict keys
(reference: models/BMCNet_plain.py), gfx950 kernels underneath."""
import torch.nn.functional as F

from .submodules import *  # noqa: F401,F403
import contextlib

from .submodules import BIE, PixelUnShuffle, initialize_weights, to_nchw, to_nhwc
from bmc_hip import ops
from bmc_hip.ops import ConvSpec, View


class Backbone(nn.Module):
    """reference: models/BMCNet_plain.py:3-33."""

    def __init__(self, n_c, n_b, scale, repeat):
        super().__init__()
        pad = (1, 1)
        s2 = scale ** 2
        if s2 % 4 or n_c % 16 or 2 * repeat > 16:
            raise NotImplementedError("bmc_hip BMCNet_plain needs an even scale, n_c a multiple of 16 and repeat <= 8")
        self.s2, self.s2p, self.cop = s2, ops.round_up(s2, 16), ops.round_up(2 * s2, 16)      # see models/BMCNet.py
        self.conv_f1 = nn.Conv2d(s2 + n_c + 2 * repeat, n_c, 3, 1, padding=pad)
        self.conv_f2 = self.conv_f1
        self.conv_fs = nn.Conv2d(s2 * 2 + n_c + 2 * 2 * repeat, n_c, 3, 1, padding=pad)
        self.para_reschunk = nn.ModuleList([BIE(n_c)] * n_b)
        self.scale = scale
        self.conv_h = nn.Conv2d(n_c, n_c, 3, 1, padding=pad)
        self.conv_o = nn.Conv2d(n_c * 2, s2 * 2, 3, 1, padding=pad)
        initialize_weights([self.conv_f1, self.conv_f2, self.conv_h, self.conv_o], 0.1)   # conv_fs keeps torch's default init, as in the reference
        r = repeat
        pad16 = lambda used: list(used) + [-1] * (16 - len(used))
        rng = lambda a, n: list(range(a, a + n))
        padk = lambda a, n: rng(a, n) + [-1] * (self.s2p - n)
        self._sp_f1 = ConvSpec([pad16(rng(0, 2 * r)), rng(2 * r, n_c), padk(2 * r + n_c, s2)])
        self._sp_fs = ConvSpec([pad16(rng(0, 2 * r)), pad16(rng(2 * r, 2 * r)), rng(4 * r, n_c), padk(4 * r + n_c, s2),
                                padk(4 * r + n_c + s2, s2)])
        self._sp_h = ConvSpec.dense(n_c)
        self._sp_o = ConvSpec.dense(n_c, n_c)

    def forward_nhwc(self, xin12, h, o12, zero_state=True):
        ops.wgrad_join()         # (a backward pass that raised leaves weight gradients running on the side stream: ops.wgrad_join)
        ops.next_window()        # (weight-gradient uses are merged within a window: ops.wgrad_wino)
        B = h.shape[0]
        # (F(4x4) or a kernel that is exact on empty receptive fields: decided per launch from its bias, ops.wino_ok's exact-zero
        #  rule; `zero_state` is unused since round 5)
        ops.prime_bias_dense(self._biases3())
        x12 = ops.conv([View(xin12), View(h, mod=B), View(o12)], self.conv_f1.weight, self.conv_f1.bias, self._sp_f1,
                       B=2 * B, relu=True)
        xs = ops.conv([View(xin12, b0=0), View(xin12, b0=B), View(h), View(o12, b0=0), View(o12, b0=B)],
                      self.conv_fs.weight, self.conv_fs.bias, self._sp_fs, B=B, relu=True)
        # (both ReLU outputs carry a positive channel at every pixel once their biases have a positive element: dense inputs for
        #  the rest of the window, ops.dense_inputs)
        dense = ops.bias_positive((self.conv_f1.bias, self.conv_fs.bias))
        with (ops.dense_inputs() if dense else contextlib.nullcontext()):
            for layer in self.para_reschunk:
                x12, xs = layer.forward_twin(x12, xs)
            x_h = ops.conv([View(xs)], self.conv_h.weight, self.conv_h.bias, self._sp_h, relu=True)
            w_o, b_o = self.conv_o.weight, self.conv_o.bias
            if self.cop != 2 * self.s2:
                w_o = F.pad(w_o, (0, 0, 0, 0, 0, 0, 0, self.cop - 2 * self.s2))
                b_o = F.pad(b_o, (0, self.cop - 2 * self.s2))
            x_o = ops.conv([View(x12, b0=0), View(x12, b0=B)], w_o, b_o, self._sp_o, B=B, cache=self.cop == 2 * self.s2)
        if self.cop != 2 * self.s2:
            x_o = x_o[..., :2 * self.s2].contiguous()
        return x_h, x_o

    def _biases3(self):
        b3 = getattr(self, "_b3", None)
        if b3 is None:
            seen = {}
            for m in self.modules():
                if isinstance(m, nn.Conv2d) and tuple(m.kernel_size) == (3, 3) and m.bias is not None:
                    seen.setdefault(id(m.bias), m.bias)
            b3 = self._b3 = list(seen.values())
        return b3

    def pad_o(self, o12):
        return o12 if self.s2p == self.s2 else F.pad(o12, (0, self.s2p - self.s2))

    def forward(self, xs, h, o):
        x1, x2 = xs
        B, r2 = x1.shape[0], x1.shape[1]
        s2 = self.scale ** 2
        z = lambda a: torch.cat([to_nhwc(a), a.new_zeros(B, a.shape[2], a.shape[3], 16 - r2)], 3)
        xin12 = torch.cat([z(x1), z(x2)], 0).contiguous()
        on = to_nhwc(o)
        o12 = self.pad_o(torch.cat([on[..., :s2], on[..., s2:]], 0).contiguous())
        return tuple(to_nchw(t) for t in self.forward_nhwc(xin12, to_nhwc(h), o12))


class BMCNet_plain(nn.Module):
    """reference: models/BMCNet_plain.py:36-68."""

    def __init__(self, scale, n_c, n_b, repeat=3):
        super().__init__()
        self.neuro = Backbone(n_c, n_b, scale, repeat=repeat)
        self.scale = scale
        self.down = PixelUnShuffle(scale)
        self.repeat = repeat

    def forward_loss(self, x, x_h, x_o, init, gt):
        """forward() + nn.MSELoss()(prediction, gt) computed by the head kernel -> (x_h, prediction, mse)."""
        return self.forward(x, x_h, x_o, init, _gt=gt)

    def forward(self, x, x_h, x_o, init, _gt=None):
        s2 = self.scale ** 2
        xin12 = ops.pack_inputs(x, self.repeat)
        if init:
            on = to_nhwc(x_o)
            o12 = torch.cat([on[..., :s2], on[..., s2:]], 0)
        else:
            o12 = ops.pixel_unshuffle_nhwc(x_o, self.scale, split=2)
        o12 = self.neuro.pad_o(o12)
        n_h, o = self.neuro.forward_nhwc(xin12, to_nhwc(x_h), o12, zero_state=bool(init))
        if _gt is not None:
            pred, mse = ops.head_mse(o, x[:, :, 1], _gt, self.scale)
            return to_nchw(n_h), pred, mse
        return to_nchw(n_h), ops.head(o, x[:, :, 1], self.scale)
