"""BMCNet on MI355X: same class names, constructor/forward signatures and
state_dict keys as the reference (models/BMCNet.py), executed by the gfx950
kernels of libbmc_hip.so.

Execution differs from the reference on purpose (results do not):
  * activations are NHWC fp32; no torch.cat ever materialises -- convolutions
    read up to six channel-slices directly;
  * the polarity twins that share weights (p / n branches, conv1 == conv2,
    convf1 == convf2, the two lBIE calls) run as ONE launch over a doubled
    batch; the three conv_fs calls run as one launch over a tripled batch;
  * ReLU, bias and residual adds live in convolution epilogues.
"""
import torch.nn.functional as F

import contextlib

from .submodules import *  # noqa: F401,F403  (same star-import surface as the reference)
from .submodules import BIE, PixelUnShuffle, ResidualBlock_noBN, initialize_weights, to_nchw, to_nhwc
from bmc_hip import bie, ops
from bmc_hip.ops import ConvSpec, View


class ParallelBlk(nn.Module):
    """reference: models/BMCNet.py:3-32."""

    def __init__(self, nf=64):
        super().__init__()
        self.conv1 = ResidualBlock_noBN(nf)
        self.conv2 = self.conv1
        self.conv1_st = ResidualBlock_noBN(nf)
        self.conv2_st = self.conv1_st
        self.lBIE = BIE(nf)   # local BIE
        self.gBIE = BIE(nf)   # global BIE
        initialize_weights([self.conv1, self.conv2, self.conv1_st, self.conv2_st], 0.1)

    def forward_nhwc(self, x12, xs, xst12, xsst12, need_st=True, first=False):
        """x12 = [x_1; x_2], xst12 = [x_1_st; x_2_st], xsst12 = [x_1_s_st; x_2_s_st] (batch-stacked twins).
        need_st=False: the caller will not read the returned xst12 (it is None then).
        first: unused since round 5 (the exact-zero rule is decided per launch from the biases: ops.wino_ok)."""
        fused = need_st or (bie.chain_supported(x12.shape[-1]))
        if fused:
            # both residual blocks write into the halves of one buffer, so the two lBIE calls (and, inside them, the
            # weight-shared conv1/conv2 and convf1/convf2 pairs) run as ONE fused twin node over 4B samples
            pair = self._res_pair(x12, xst12)
        else:
            x12 = self.conv1.forward_nhwc(x12)
            xst12 = self.conv1_st.forward_nhwc(xst12)
        if need_st:
            o, xsst12 = self.lBIE.forward_twin(pair, xsst12)
            x12, xst12 = bie.Unstack2Fn.unstack(o)
        elif fused:
            # the same with only the first output of the local BIE (one fused node again: BIE.forward_first)
            x12, xsst12 = self.lBIE.forward_first(pair, xsst12)
            xst12 = None
        else:
            x12, xst12, xsst12 = self.lBIE.forward_pair(x12, xst12, xsst12, need_second=False)
        x12, xs = self.gBIE.forward_twin(x12, xs)
        return x12, xs, xst12, xsst12

    def _res_pair(self, x12, xst12):
        """[conv1(x12); conv1_st(xst12)] as one tensor [2 * B2, H, W, C] (the two residual blocks have their own weights)."""
        B2, H, W, _ = x12.shape
        if ops.pair_small(B2, H, W):
            # small frames: each convolution of the two blocks as ONE two-group launch over the stacked inputs (ops.pair_small)
            xx = torch.cat([x12, xst12], 0)
            c1, c2 = self.conv1, self.conv1_st
            t = ops.conv_groups([View(xx)], (c1.conv1.weight, c2.conv1.weight), (c1.conv1.bias, c2.conv1.bias), c1._spec, relu=True)
            return ops.conv_groups([View(t)], (c1.conv2.weight, c2.conv2.weight), (c1.conv2.bias, c2.conv2.bias), c1._spec,
                                   residual=View(xx))
        slot = ops.OutSlot(torch.empty((2 * B2,) + tuple(x12.shape[1:]), device=x12.device, dtype=x12.dtype), 0)
        a = self.conv1.forward_nhwc(x12, out=slot)
        b = self.conv1_st.forward_nhwc(xst12, out=ops.OutSlot(slot.t, B2))
        return bie.Stack2Fn.apply(a, b, slot)

    def forward(self, x_1, x_2, x_s, x_1_st, x_2_st, x_1_s_st, x_2_s_st):
        B = x_1.shape[0]
        st = lambda a, b: torch.cat([to_nhwc(a), to_nhwc(b)], 0)
        x12, xs, xst12, xsst12 = self.forward_nhwc(st(x_1, x_2), to_nhwc(x_s), st(x_1_st, x_2_st),
                                                   st(x_1_s_st, x_2_s_st))
        n = to_nchw
        return n(x12[:B]), n(x12[B:]), n(xs), n(xst12[:B]), n(xst12[B:]), n(xsst12[:B]), n(xsst12[B:])


class Backbone(nn.Module):
    """reference: models/BMCNet.py:35-84."""

    def __init__(self, n_c, n_b, scale, repeat):
        super().__init__()
        pad = (1, 1)
        s2 = scale ** 2
        if s2 % 4 or n_c % 16 or 2 * repeat > 16:
            raise NotImplementedError("bmc_hip BMCNet needs an even scale, n_c a multiple of 16 and repeat <= 8 "
                                      "(scale=%d, n_c=%d, repeat=%d)" % (scale, n_c, repeat))
        # the kernels read operands in 16-channel granules: the s^2 sub-pixel channels of each polarity half of o (4 at
        # x2 SR) are carried in tensors padded to s2p channels, and conv_o's 2 s^2 outputs in a tensor padded to cop
        self.s2, self.s2p, self.cop = s2, ops.round_up(s2, 16), ops.round_up(2 * s2, 16)
        self.conv_fpst = nn.Conv2d(s2 + n_c + 2 * repeat, n_c, 3, 1, padding=pad)
        self.conv_fnst = self.conv_fpst
        self.conv_fps = nn.Conv2d(repeat + n_c, n_c, 3, 1, padding=pad)
        self.conv_fns = self.conv_fps
        self.conv_fs = nn.Conv2d(s2 * 2 + n_c * 3, n_c, 3, 1, padding=pad)
        self.para_reschunk = nn.ModuleList([ParallelBlk(n_c)] * n_b)
        self.scale = scale
        self.conv_hs = nn.Conv2d(n_c, n_c, 3, 1, padding=pad)
        self.conv_hp = nn.Conv2d(n_c, n_c, 3, 1, padding=pad)
        self.conv_hn = nn.Conv2d(n_c, n_c, 3, 1, padding=pad)
        self.conv_o = nn.Conv2d(n_c * 2, s2 * 2, 3, 1, padding=pad)
        initialize_weights([self.conv_fpst, self.conv_fnst, self.conv_fps, self.conv_fns, self.conv_fs, self.conv_hs,
                            self.conv_hp, self.conv_hn, self.conv_o], 0.1)
        r = repeat
        pad16 = lambda used: list(used) + [-1] * (16 - len(used))
        rng = lambda a, n: list(range(a, a + n))
        # packed-K layouts (reference concat orders: models/BMCNet.py:60-73,78-82)
        padk = lambda a, n: rng(a, n) + [-1] * (self.s2p - n)
        self._sp_fpst = ConvSpec([pad16(rng(0, 2 * r)), rng(2 * r, n_c), padk(2 * r + n_c, s2)])
        self._sp_fps = ConvSpec([pad16([-1] * r + rng(0, r)), rng(r, n_c)])
        # conv_fs is applied three times to cat[xp_st, xn_st, h*, o] with only h* changing (models/BMCNet.py:70-73):
        # the contribution of the shared 2*n_c + 2*s2 input channels is computed once (with the bias), the h* part per call
        # (both launches take the PARAMETER conv_fs.weight and name their columns of it: no per-window slice / cat copies,
        # the packs are cached per optimizer step and both weight gradients add straight into conv_fs.weight.grad)
        cin_fs = 3 * n_c + 2 * s2
        self._sp_fs_shared = ConvSpec([rng(0, n_c), rng(n_c, n_c), padk(3 * n_c, s2), padk(3 * n_c + s2, s2)], cin=cin_fs)
        self._sp_fs_h = ConvSpec([rng(2 * n_c, n_c)], cin=cin_fs)
        self._sp_h = ConvSpec.dense(n_c)
        self._sp_o = ConvSpec.dense(n_c, n_c)
        self.n_c = n_c

    def forward_nhwc(self, xin12, h3, o12, zero_state=True):
        """xin12 [2B,H,W,16]: packed polarity inputs (p batch-half, n batch-half);
        h3 [3B,H,W,n_c] = [hp; hn; hs]; o12 [2B,H,W,s^2] = [o[:, :s^2]; o[:, s^2:]] (channel halves batch-stacked).
        Returns x_h, x_h_p, x_h_n, x_o (NHWC)."""
        ops.wgrad_join()         # (a backward pass that raised leaves weight gradients running on the side stream: ops.wgrad_join)
        ops.next_window()        # (weight-gradient uses are merged within a window: ops.wgrad_wino)
        B = o12.shape[0] // 2
        hpn = h3[:2 * B]
        # (which 3x3 launches may take the F(4x4) kernel: the exact-zero rule above ops.wino_ok; round 4 keyed it on `zero_state`,
        #  which a sparse recording defeats.  The input-fusion convolutions read raw event counts and a state that may be zero:
        #  decided per launch from their own biases.  Once each of them has a positive bias element, every pixel of st12, s12, fs3
        #  carries a positive channel and the residual blocks / BIEs keep it non-zero: the rest of the window has dense inputs.
        #  One stack of device reductions + one host read per optimizer step for all 3x3 biases of the backbone:)
        ops.prime_bias_dense(self._biases3())
        st12, s12, sst12, xs = self._input_fusion(xin12, h3, hpn, o12, B)
        dense = ops.bias_positive((self.conv_fpst.bias, self.conv_fps.bias, self.conv_fs.bias))
        with (ops.dense_inputs() if dense else contextlib.nullcontext()):
            n_layers = len(self.para_reschunk)
            for i, layer in enumerate(self.para_reschunk):      # x*_st of the last block is never read: skip what only feeds it
                s12, xs, st12, sst12 = layer.forward_nhwc(s12, xs, st12, sst12, need_st=i + 1 < n_layers)
            return self._tail(s12, xs, sst12, B)

    def _biases3(self):
        """The bias vectors of the backbone's 3x3 convolutions (aliases once): what ops.wino_ok's exact-zero rule reads."""
        b3 = getattr(self, "_b3", None)
        if b3 is None:
            seen = {}
            for m in self.modules():
                if isinstance(m, nn.Conv2d) and tuple(m.kernel_size) == (3, 3) and m.bias is not None:
                    seen.setdefault(id(m.bias), m.bias)
            b3 = self._b3 = list(seen.values())
        return b3

    def _input_fusion(self, xin12, h3, hpn, o12, B):
        st12 = ops.conv([View(xin12), View(hpn), View(o12)], self.conv_fpst.weight, self.conv_fpst.bias, self._sp_fpst,
                        relu=True)                                         # [xp_st; xn_st]
        s12 = ops.conv([View(xin12), View(hpn)], self.conv_fps.weight, self.conv_fps.bias, self._sp_fps,
                       relu=True)                                          # [xp_s; xn_s]
        # conv_fs on cat[xp_st, xn_st, h*, o] for h* = hp, hn, hs: one launch over 3B
        wfs = self.conv_fs.weight
        shared = ops.conv([View(st12, b0=0), View(st12, b0=B), View(o12, b0=0), View(o12, b0=B)], wfs,
                          self.conv_fs.bias, self._sp_fs_shared, B=B)      # input channels of xp_st, xn_st, o (+ the bias)
        fs3 = ops.conv([View(h3)], wfs, None, self._sp_fs_h, B=3 * B, relu=True,
                       residual=View(shared, mod=B), rule=self.conv_fs.bias)                       # + those of h*: [xs_p_st; xs_n_st; xs]
        sst12, xs = bie.Unstack2Fn.apply(fs3, 2 * B)       # (one concatenation in backward instead of two zero-filled slice gradients + add)
        return st12, s12, sst12, xs

    def _tail(self, s12, xs, sst12, B):
        # the three new states go into ONE buffer [x_h; x_h_p; x_h_n]: the next window reads them as h3 without a copy
        # (ops.stack_states recognises the adjacent views)
        hbuf = torch.empty((3 * B,) + tuple(xs.shape[1:]), device=xs.device, dtype=xs.dtype)
        x_h = ops.conv([View(xs)], self.conv_hs.weight, self.conv_hs.bias, self._sp_h, relu=True, out=ops.OutSlot(hbuf, 0))
        x_hpn = ops.conv_groups([View(sst12)], (self.conv_hp.weight, self.conv_hn.weight), (self.conv_hp.bias, self.conv_hn.bias),
                                self._sp_h, relu=True, out=ops.OutSlot(hbuf, B))
        x_hp, x_hn = bie.Unstack2Fn.apply(x_hpn)
        w_o, b_o = self.conv_o.weight, self.conv_o.bias
        if self.cop != 2 * self.s2:        # x2 SR: 8 output channels -> computed as 16 (zero rows), the head reads the first 8
            w_o = F.pad(w_o, (0, 0, 0, 0, 0, 0, 0, self.cop - 2 * self.s2))
            b_o = F.pad(b_o, (0, self.cop - 2 * self.s2))
        x_o = ops.conv([View(s12, b0=0), View(s12, b0=B)], w_o, b_o, self._sp_o, B=B, cache=self.cop == 2 * self.s2)
        if self.cop != 2 * self.s2:
            x_o = x_o[..., :2 * self.s2].contiguous()
        return x_h, x_hp, x_hn, x_o

    def forward(self, xs, hp, hn, hs, o):
        """NCHW interface of the reference (xs = [x1p, x1n, x2p, x2n], 3 repeated channels each)."""
        x1p, x1n, x2p, x2n = xs
        B, r = x1p.shape[0], x1p.shape[1]
        s2 = self.scale ** 2
        z = lambda a, b: torch.cat([to_nhwc(a), to_nhwc(b), a.new_zeros(B, a.shape[2], a.shape[3], 16 - 2 * r)], 3)
        xin12 = torch.cat([z(x1p, x2p), z(x1n, x2n)], 0).contiguous()
        h3 = torch.cat([to_nhwc(hp), to_nhwc(hn), to_nhwc(hs)], 0)
        on = to_nhwc(o)
        o12 = self.pad_o(torch.cat([on[..., :s2], on[..., s2:]], 0).contiguous())
        return tuple(to_nchw(t) for t in self.forward_nhwc(xin12, h3, o12))

    def pad_o(self, o12):
        """[2B,H,W,s^2] -> [2B,H,W,s2p] (zero channels up to the 16-channel granule; a no-op for x4 / x8 SR)."""
        return o12 if self.s2p == self.s2 else F.pad(o12, (0, self.s2p - self.s2))


class BMCNet(nn.Module):
    """reference: models/BMCNet.py:87-121."""

    def __init__(self, scale, n_c, n_b, repeat=3):
        super().__init__()
        self.neuro = Backbone(n_c, n_b, scale, repeat=repeat)
        self.scale = scale
        self.down = PixelUnShuffle(scale)
        self.repeat = repeat

    def forward_loss(self, x, x_h, x_h_p, x_h_n, x_o, init, gt):
        """forward() + nn.MSELoss()(prediction, gt) with the loss computed by the head kernel (one pass over the HR
        tensor forward, the loss gradient folded into the head's backward): -> (x_h, x_h_p, x_h_n, prediction, mse).
        gt must have the prediction's size (otherwise use forward() + bicubic resize, train.py:227-231)."""
        return self.forward(x, x_h, x_h_p, x_h_n, x_o, init, _gt=gt)

    def forward(self, x, x_h, x_h_p, x_h_n, x_o, init, _gt=None):
        """x [B,2,T>=2,H,W]; x_h/x_h_p/x_h_n [B,n_c,H,W]; x_o [B,2*s*s,H,W] if init else the previous HR
        prediction [B,2,sH,sW]; returns (x_h, x_h_p, x_h_n, prediction [B,2,sH,sW])."""
        B = x.shape[0]
        s2 = self.scale ** 2
        xin12 = ops.pack_inputs(x, self.repeat)
        if init:
            on = to_nhwc(x_o)
            o12 = torch.cat([on[..., :s2], on[..., s2:]], 0)
        else:       # the previous HR prediction, unshuffled straight into the batch-stacked channel halves
            o12 = ops.pixel_unshuffle_nhwc(x_o, self.scale, split=2)
        o12 = self.neuro.pad_o(o12)
        # the reference passes (x_h, x_h_p, x_h_n) positionally into Backbone.forward(xs, hp, hn, hs, o)
        # (models/BMCNet.py:115,118 vs :57): x_h acts as hp, x_h_p as hn, x_h_n as hs.
        h3 = ops.stack_states([x_h, x_h_p, x_h_n])
        n_h, n_hp, n_hn, o = self.neuro.forward_nhwc(xin12, h3, o12, zero_state=bool(init))
        if _gt is not None:
            pred, mse = ops.head_mse(o, x[:, :, 1], _gt, self.scale)
            return to_nchw(n_h), to_nchw(n_hp), to_nchw(n_hn), pred, mse
        pred = ops.head(o, x[:, :, 1], self.scale)
        return to_nchw(n_h), to_nchw(n_hp), to_nchw(n_hn), pred
