"""Fused BIE block (bilateral information exchange, reference models/submodules.py:38-77) in its "twin" form:
`first` and `second` are the two batch halves of ONE tensor x12, so every weight-shared pair of the reference
(conv1 == conv2, convf1 == convf2) is one launch over the doubled batch.

Forward and backward are written out launch by launch (no autograd inside): every gradient that has several
contributions (x12: residual block + convf + value conv; xs: convf twice + skip; center: Gram + unclustering; v: Gram
+ attention) is accumulated by convolution epilogues (residual / accumulate) instead of separate add kernels, the
ReLU backward of the residual block is a mask epilogue, and the crossed skip connections (out_1 + Res(x_2),
out_2 + Res(x_1)) are batch-rotated operand reads -- no roll / cat copies.
"""
from __future__ import annotations

import torch

from . import lib
from .ops import (CK, ConvSpec, _dense_spec, _need_gpu, _packed_weight, _packed_weight_t, _src, _stream, conv_raw,
                  coutpad, pgemm_raw, round_up)

_SPEC2 = {}


def _spec2(c):
    s = _SPEC2.get(c)
    if s is None:
        s = ConvSpec.dense(c, c)
        _SPEC2[c] = s
    return s


def _conv(srcs, w4, spec, owner, bias, out, B, relu=False, residual=None, mask=None, bpg=None, accumulate=False,
          out_b0=0):
    """Forward-style launch: out[out_b0 : out_b0+B] = epi(conv(cat(srcs)) + bias)."""
    G, Cout, Cin, taps = w4.shape
    _, H, W, Co = out.shape
    wp = _packed_weight(w4, spec, owner)
    conv_raw(srcs, wp, spec.kpad * taps * coutpad(Cout), bias, Cout if bias is not None else 0,
             out.data_ptr() + 4 * out_b0 * H * W * Co, H * W * Co, Co, B, H, W, Cout, taps, relu=relu, residual=residual,
             bpg=bpg, accumulate=accumulate, mask=mask, flops=2.0 * B * H * W * Cout * taps * spec.cin)


def _dgrad(g_src, w4, spec, src_index, owner, out, B, residual=None, mask=None, bpg=None, accumulate=False, out_b0=0):
    """Data gradient w.r.t. source `src_index` of the conv with weights w4: out[out_b0:+B] (=|+=) conv^T(g)."""
    G, Cout, Cin, taps = w4.shape
    _, H, W, Co = out.shape
    nch = spec.nch[src_index]
    wt = _packed_weight_t(w4, spec, src_index, owner)
    conv_raw([g_src], wt, round_up(Cout, CK) * taps * coutpad(nch), None, 0, out.data_ptr() + 4 * out_b0 * H * W * Co,
             H * W * Co, Co, B, H, W, nch, taps, residual=residual, mask=mask, bpg=bpg, accumulate=accumulate,
             flops=2.0 * B * H * W * spec.real_nch[src_index] * taps * Cout)


def _wgrad(a_src, x_srcs, spec, B, H, W, taps, Cout, dev, G=1):
    """-> (dW flat [G*Cout*Cin*taps], db [G,Cout]): weight gradient + bias gradient (column sums of the same A operand,
    taken from the tiles the pixel-reduction GEMM stages anyway)."""
    slabs, nsplit, _, bsl = pgemm_raw(a_src, x_srcs, B, H, W, taps, B // G, Cout, spec.kpad, dev,
                                      flops=2.0 * B * H * W * Cout * taps * spec.cin, want_bias=True)
    dw = torch.empty(G * Cout * spec.cin * taps, device=dev, dtype=torch.float32)
    db = torch.empty((G, Cout), device=dev, dtype=torch.float32)
    lib.call(lib._red_w, "bmc_pgemm_reduce_weight", slabs.data_ptr(), nsplit, G, taps, Cout, spec.kpad,
             spec.kmap(dev).data_ptr(), spec.cin, dw.data_ptr(), 0, bsl.data_ptr(), db.data_ptr(), _stream())
    return dw, db


class BIETwinFn(torch.autograd.Function):
    """inputs: x12 [2n,H,W,C] = [first; second], xs [n,H,W,C], then the 16 parameter tensors
    (res.conv1 w,b; res.conv2 w,b; convf w,b; norm w,b; clustering w,b; unclustering w,b; v1 w,b; v2 w,b).
    outputs: o12 = [softmax(att1) v1 + Res(second); softmax(att2) v2 + Res(first)], xs_new."""

    @staticmethod
    def forward(ctx, x12, xs, rw1, rb1, rw2, rb2, wf, bf, gamma, beta, wc, bc, wu, bu, wv1, bv1, wv2, bv2, scale, eps):
        _need_gpu(x12)
        x12, xs = x12.contiguous(), xs.contiguous()
        B2, H, W, Cn = x12.shape
        n = B2 // 2
        dev = x12.device
        s1, s2 = _dense_spec(Cn), _spec2(Cn)
        new = lambda b: torch.empty((b, H, W, Cn), device=dev, dtype=torch.float32)
        X = lambda t, **k: _src(t, 0, Cn, k.get("shift", 0), k.get("mod"), k.get("b0", 0), k.get("B", t.shape[0]))
        d = lambda t: t.detach()
        # residual block on both halves (shared weights)
        t12, r12 = new(B2), new(B2)
        _conv([X(x12)], d(rw1).reshape(1, Cn, Cn, 9), s1, rw1, d(rb1), t12, B2, relu=True)
        _conv([X(t12)], d(rw2).reshape(1, Cn, Cn, 9), s1, rw2, d(rb2), r12, B2, residual=X(x12))
        # centres: clustering(LN(convf(cat[xs, other half])))
        z12, y12, c12 = new(B2), new(B2), new(B2)
        _conv([X(xs, mod=n, B=B2), X(x12, shift=n, mod=B2)], d(wf).reshape(1, Cn, 2 * Cn, 1), s2, wf, d(bf), z12, B2)
        stats = torch.empty(B2 * H * W * 2, device=dev, dtype=torch.float32)
        lib.call(lib._ln_fwd, "bmc_layernorm_fwd", z12.data_ptr(), gamma.data_ptr(), beta.data_ptr(), B2 * H * W, Cn, eps,
                 y12.data_ptr(), stats.data_ptr(), _stream())
        _conv([X(y12)], d(wc).reshape(1, Cn, Cn, 1), s1, wc, d(bc), c12, B2)
        # values: v1 on the first half, v2 on the second (two weight groups)
        v12 = new(B2)
        wv = torch.stack([d(wv1).reshape(Cn, Cn, 1), d(wv2).reshape(Cn, Cn, 1)])
        bv = torch.stack([d(bv1), d(bv2)])
        _conv([X(x12)], wv, s1, None, bv, v12, B2, bpg=n)
        # channel attention per sample
        slabs, nsplit, G = pgemm_raw(X(c12), [X(v12)], B2, H, W, 1, 1, Cn, Cn, dev, flops=2.0 * B2 * H * W * Cn * Cn)
        att = torch.empty((B2, Cn, Cn), device=dev, dtype=torch.float32)
        lib.call(lib._red_p, "bmc_pgemm_reduce_plain", slabs.data_ptr(), nsplit, G, Cn, Cn, scale, att.data_ptr(), _stream())
        p = torch.empty_like(att)
        lib.call(lib._sm_fwd, "bmc_softmax_fwd", att.data_ptr(), B2 * Cn, Cn, p.data_ptr(), _stream())
        o12 = new(B2)
        _conv([X(v12)], p.view(B2, Cn, Cn, 1), s1, None, None, o12, B2, residual=X(r12, shift=n, mod=B2), bpg=1)
        # shared stream: unclustering(cat[c1, c2]) + xs
        xs_new = new(n)
        _conv([X(c12, b0=0, B=n), X(c12, b0=n, B=n)], d(wu).reshape(1, Cn, 2 * Cn, 1), s2, wu, d(bu), xs_new, n, residual=X(xs))
        ctx.save_for_backward(x12, xs, t12, z12, stats, y12, c12, v12, p, rw1, rw2, wf, gamma, wc, wu, wv1, wv2)
        ctx.owners = (rw1, rw2, wf, wc, wu)
        ctx.scale = scale
        return o12, xs_new

    @staticmethod
    def backward(ctx, do12, dxs_new):
        x12, xs, t12, z12, stats, y12, c12, v12, p, rw1, rw2, wf, gamma, wc, wu, wv1, wv2 = ctx.saved_tensors
        o_rw1, o_rw2, o_wf, o_wc, o_wu = ctx.owners
        B2, H, W, Cn = x12.shape
        n = B2 // 2
        dev = x12.device
        s1, s2 = _dense_spec(Cn), _spec2(Cn)
        new = lambda b: torch.empty((b, H, W, Cn), device=dev, dtype=torch.float32)
        X = lambda t, **k: _src(t, 0, Cn, k.get("shift", 0), k.get("mod"), k.get("b0", 0), k.get("B", t.shape[0]))
        g_o = do12.contiguous() if do12 is not None else torch.zeros_like(x12)
        g_x = dxs_new.contiguous() if dxs_new is not None else torch.zeros_like(xs)
        w_r1, w_r2 = rw1.detach().reshape(1, Cn, Cn, 9), rw2.detach().reshape(1, Cn, Cn, 9)
        w_f, w_c, w_u = (rw.detach().reshape(1, Cn, k, 1) for rw, k in ((wf, 2 * Cn), (wc, Cn), (wu, 2 * Cn)))
        w_v = torch.stack([wv1.detach().reshape(Cn, Cn, 1), wv2.detach().reshape(Cn, Cn, 1)])

        # ---- out = P v (+ rotated residual): dP, dv
        slabs, nsplit, G = pgemm_raw(X(g_o), [X(v12)], B2, H, W, 1, 1, Cn, Cn, dev, flops=2.0 * B2 * H * W * Cn * Cn)
        dp = torch.empty_like(p)
        lib.call(lib._red_p, "bmc_pgemm_reduce_plain", slabs.data_ptr(), nsplit, G, Cn, Cn, 1.0, dp.data_ptr(), _stream())
        dv12 = new(B2)
        _conv([X(g_o)], p.transpose(1, 2).contiguous().view(B2, Cn, Cn, 1), s1, None, None, dv12, B2, bpg=1)
        # ---- softmax, Gram (att = scale * center v^T)
        da = torch.empty_like(p)
        lib.call(lib._sm_bwd, "bmc_softmax_bwd", p.data_ptr(), dp.data_ptr(), B2 * Cn, Cn, ctx.scale, da.data_ptr(), _stream())
        dc12 = new(B2)
        _conv([X(v12)], da.view(B2, Cn, Cn, 1), s1, None, None, dc12, B2, bpg=1)                         # d center
        _conv([X(c12)], da.transpose(1, 2).contiguous().view(B2, Cn, Cn, 1), s1, None, None, dv12, B2, bpg=1,
              accumulate=True)                                                                           # dv +=
        # ---- unclustering(cat[c1, c2]) + xs
        dwu, dbu = _wgrad(X(g_x), [X(c12, b0=0, B=n), X(c12, b0=n, B=n)], s2, n, H, W, 1, Cn, dev)
        _dgrad(X(g_x), w_u, s2, 0, o_wu, dc12, n, accumulate=True, out_b0=0)
        _dgrad(X(g_x), w_u, s2, 1, o_wu, dc12, n, accumulate=True, out_b0=n)
        # ---- value convs (two weight groups)
        dwv, dbv = _wgrad(X(dv12), [X(x12)], s1, B2, H, W, 1, Cn, dev, G=2)
        dwv = dwv.view(2, Cn, Cn, 1, 1)
        dx12 = new(B2)
        _dgrad(X(dv12), w_v, s1, 0, None, dx12, B2, bpg=n)                                               # dx12  =
        # ---- clustering, LayerNorm, convf
        dwc, dbc = _wgrad(X(dc12), [X(y12)], s1, B2, H, W, 1, Cn, dev)
        dy12 = new(B2)
        _dgrad(X(dc12), w_c, s1, 0, o_wc, dy12, B2)
        dz12 = new(B2)
        ws = torch.empty(2 * 1024 * Cn, device=dev, dtype=torch.float32)
        dgamma = torch.empty(Cn, device=dev, dtype=torch.float32)
        dbeta = torch.empty(Cn, device=dev, dtype=torch.float32)
        lib.call(lib._ln_bwd, "bmc_layernorm_bwd", dy12.data_ptr(), z12.data_ptr(), stats.data_ptr(), gamma.data_ptr(),
                 B2 * H * W, Cn, dz12.data_ptr(), ws.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), 0, _stream())
        dwf, dbf = _wgrad(X(dz12), [X(xs, mod=n, B=B2), X(x12, shift=n, mod=B2)], s2, B2, H, W, 1, Cn, dev)
        dxs = new(n)
        _dgrad(X(dz12, b0=0, B=n), w_f, s2, 0, o_wf, dxs, n, residual=X(g_x))                            # dxs  = skip + half 0
        _dgrad(X(dz12, b0=n, B=n), w_f, s2, 0, o_wf, dxs, n, accumulate=True)                            # dxs += half 1
        _dgrad(X(dz12, shift=n, mod=B2), w_f, s2, 1, o_wf, dx12, B2, accumulate=True)                    # dx12 += (rotated)
        # ---- residual block, upstream gradient = batch-rotated g_o
        g_r = X(g_o, shift=n, mod=B2)
        dw2, db2 = _wgrad(g_r, [X(t12)], s1, B2, H, W, 9, Cn, dev)
        dt = new(B2)
        _dgrad(g_r, w_r2, s1, 0, o_rw2, dt, B2, mask=X(t12))
        dw1, db1 = _wgrad(X(dt), [X(x12)], s1, B2, H, W, 9, Cn, dev)
        _dgrad(X(dt), w_r1, s1, 0, o_rw1, dx12, B2, residual=g_r, accumulate=True)                       # dx12 += conv1^T + skip
        return (dx12, dxs, dw1.view(rw1.shape), db1[0], dw2.view(rw2.shape), db2[0], dwf.view(wf.shape), dbf[0], dgamma, dbeta,
                dwc.view(wc.shape), dbc[0], dwu.view(wu.shape), dbu[0], dwv[0].reshape(wv1.shape), dbv[0],
                dwv[1].reshape(wv2.shape), dbv[1], None, None)


def bie_twin(m, x12, xs):
    """m: models.submodules.BIE module (parameter container)."""
    r = m.conv1
    return BIETwinFn.apply(x12, xs, r.conv1.weight, r.conv1.bias, r.conv2.weight, r.conv2.bias, m.convf1.weight,
                           m.convf1.bias, m.norm_s.weight, m.norm_s.bias, m.clustering.weight, m.clustering.bias,
                           m.unclustering.weight, m.unclustering.bias, m.v1.weight, m.v1.bias, m.v2.weight, m.v2.bias,
                           m.scale, m.norm_s.eps)


class Stack2Fn(torch.autograd.Function):
    """Declares that tensors a and b (which already ARE the two batch halves of `buf`) form one tensor -- no copy;
    backward hands each producer its half of the gradient as a view."""

    @staticmethod
    def forward(ctx, a, b, slot):
        buf, n = slot.t, a.shape[0]
        assert a.data_ptr() == buf.data_ptr() and b.data_ptr() == buf[n:].data_ptr() and buf.shape[0] == 2 * n
        ctx.n = n
        return buf

    @staticmethod
    def backward(ctx, g):
        return g[:ctx.n], g[ctx.n:], None


class Unstack2Fn(torch.autograd.Function):
    """t [2n,...] -> (t[:n], t[n:]) as views; backward concatenates the two gradients with ONE copy (the generic
    slice backward would zero-fill and add two full-size tensors)."""

    @staticmethod
    def forward(ctx, t):
        n = t.shape[0] // 2
        ctx.n = n
        ctx.shape = t.shape
        return t[:n], t[n:]

    @staticmethod
    def backward(ctx, g1, g2):
        z = lambda: torch.zeros((ctx.n,) + tuple(ctx.shape[1:]), device=(g1 if g1 is not None else g2).device)
        return torch.cat([g1 if g1 is not None else z(), g2 if g2 is not None else z()], 0)
