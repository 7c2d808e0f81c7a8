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

import ctypes as C
import os
import weakref

import torch

from . import lib, ops
from .ops import (CK, ConvSpec, _dense_spec, _need_gpu, _null_src, _packed_weight, _packed_weight_t, _src, _stream,
                  conv_raw, coutpad, pgemm_raw, round_up)

_SPEC2 = {}

# Fused centre chain (csrc/chain.hip): convf -> LayerNorm2d -> clustering in one launch per direction.  fp32 MFMA only
# (the bf16-plane modes keep the unfused launches); BMC_FUSE_CHAIN=0 switches it off (A/B measurements, cross-checks).
FUSE_CHAIN = os.environ.get("BMC_FUSE_CHAIN", "1") != "0"
_CHAIN_CACHE = {}



# Attention without the value tensor (fp32 arithmetic modes).  v = W_v x + b_v enters the BIE twice -- the Gram matrix
# att = scale * c^T v and the product out = softmax(att) v -- and both are linear in v, so with G0 = c^T x (the same
# pixel-reduction GEMM, on x instead of v) and s = the column sums of c (its bias slabs):
#     att = scale * (G0 W_v^T + s b_v^T),        out = (P W_v) x + P b_v           (P = softmax(att), per sample)
# i.e. v is never formed: the forward loses the value convolution, the backward loses its data-gradient convolution and its
# weight-gradient GEMM (dW_v = P^T dM + da^T G0 from C x C matrices, dM = g_out^T x taking the place of g_out^T v), three of
# the BIE's thirteen full-size 1x1 launches -- against a handful of C x C x C products per sample.  Same mathematics, another
# order of summation (the parity bars of tests/parity_bars.py hold it); BMC_BIE_VFREE=0 restores the explicit form, which the
# bf16 mode keeps (its operand-rounding oracle rounds v).
# Not for small launches: the three full-size launches it removes cost 12-15 us each at 31x56, the eight small ones it adds
# 5-6 us each plus their weight packs (31x56: 82.3 -> 96.6 ms, 48x64: 111.5 -> 124.9 ms; break-even at 64x96, 154.2 vs 155.0 ms;
# 90x120: 232.4 -> 228.3 ms; 180x240: 743 -> 710 ms) -- from 2^16 pixels per launch.
VFREE = os.environ.get("BMC_BIE_VFREE", "1") != "0"
VFREE_MIN_PIXELS = int(os.environ.get("BMC_BIE_VFREE_MIN_PIXELS", 1 << 16))


def vfree_supported(npx):
    return VFREE and ops.MATH in (0, 3) and npx >= VFREE_MIN_PIXELS


def _gram_on_x(c_src, x_src, B, H, W, Cn, dev):
    """-> (G0 [B,C,C], s [B,C]): G0_b = sum_px c_b[px,:]^T x_b[px,:], s_b = sum_px c_b[px,:] (one pixel-reduction launch per
    sample group + its reduction; c_src / x_src: lib.Src over B launch batches)."""
    slabs, nsplit, G, bsl = pgemm_raw(c_src, [x_src], B, H, W, 1, 1, Cn, Cn, dev, flops=2.0 * B * H * W * Cn * Cn, want_bias=True)
    G0 = torch.empty((B, Cn, Cn), device=dev, dtype=torch.float32)
    sc = torch.empty((B, Cn), device=dev, dtype=torch.float32)
    lib.call(lib._red_w, "bmc_pgemm_reduce_weight", slabs.data_ptr(), nsplit, G, 1, Cn, Cn, None, Cn, G0.data_ptr(), 0,
             bsl.data_ptr(), sc.data_ptr(), _stream())
    return G0, sc


def _times_wt(m, vec, wg, bg, bpg, alpha, out):
    """out[b] = alpha (m[b] W_g^T + vec[b] b_g^T)  (m [B,C,C], vec [B,C]; wg [G,C(j),C(k)], bg [G,C]; g = b // bpg): the Gram
    matrix of the attention from G0 = center^T x, and dP from dM = g_out^T x."""
    B, Cn, _ = m.shape
    ops.small_mm([((m, Cn * Cn, 0, Cn, 1), (wg, 0, Cn * Cn, 1, Cn), None)], B, bpg, Cn, Cn, Cn, c=(out, Cn * Cn, 0, Cn, 1),
                 alpha=alpha, uv=((vec, Cn, 0), (bg, 0, Cn)))
    return out


def _times_w(m, wg, bg, bpg, out, out_strides, vec=None):
    """out[b][i][k] = sum_j m[b][i][j] W_g[j][k] (written with out_strides = (batch, i, k) strides: plain, into a column block
    of a wider matrix, or transposed), vec[b][i] = sum_j m[b][i][j] b_g[j]."""
    B, Cn, _ = m.shape
    ops.small_mm([((m, Cn * Cn, 0, Cn, 1), (wg, 0, Cn * Cn, Cn, 1), (bg, 0, Cn, 1) if vec is not None else None)], B, bpg, Cn, Cn, Cn,
                 c=(out, out_strides[0], 0, out_strides[1], out_strides[2]), vec_out=(vec, Cn, 0) if vec is not None else None)


def _value_param_grads(p, dM, tg, da, G0, sc, w_params, b_params, npx):
    """dW_v[g] = sum over the group's samples of P^T dM + da^T G0, db_v[g] = sum of P^T t + da^T s.  The per-sample products are
    written in the slab layout of the pixel-reduction GEMM (sample = split), so the sum over a group's samples, the routing
    into the parameters' .grad (or back to autograd) and the weight-gradient stream are ops.reduce_wgrad's, as for every other
    weight gradient.  -> (dw [G,C,C], db [G,C]) or (None, None)."""
    groups = len(w_params)
    B, Cn, _ = p.shape
    n, cc, dev = B // groups, Cn * Cn, p.device
    mp = round_up(Cn, 32)
    slab, bslab = groups * mp * mp, groups * 4 * mp
    slabs = torch.empty(n * slab, device=dev, dtype=torch.float32)       # [n][G][1][mp][mp]; the reduction reads rows / columns < C only
    bsl = torch.zeros(n * bslab, device=dev, dtype=torch.float32)        # [n][G][4][mp]: part 0 written, parts 1-3 zero
    params = list(w_params) + list(b_params)
    with ops.wgrad_side(npx, params, (p, dM, tg, da, G0, sc)):
        ops._on_side(slabs, bsl)
        # batch b = g * n + s  ->  slab s, group g
        ops.small_mm([((p, cc, 0, 1, Cn), (dM, cc, 0, Cn, 1), (tg, Cn, 0, 1)), ((da, cc, 0, 1, Cn), (G0, cc, 0, Cn, 1), (sc, Cn, 0, 1))],
                     B, n, Cn, Cn, Cn, c=(slabs, slab, mp * mp - n * slab, mp, 1), vec_out=(bsl, bslab, 4 * mp - n * bslab))
        one = groups == 1
        return ops.reduce_wgrad(slabs, n, groups, 1, Cn, _dense_spec(Cn), dev, bsl, w_params[0] if one else tuple(w_params),
                                b_params[0] if one else tuple(b_params), (Cn, Cn) if one else (groups, Cn, Cn))


def chain_supported(Cn):
    # (bf16x6 is an fp32-equivalent mode: the native fp32 fused chain is a valid member of it)
    return FUSE_CHAIN and ops.MATH in (0, 3) and Cn in (32, 64, 128)


def _chain_streams(wf, wc, Cn):
    """Weight streams of bmc_chain_fwd / bmc_chain_bwd for (convf.weight, clustering.weight), cached per parameter
    version: the forward stream is pack(W_f) | pack(W_c); the backward stream is T(W_c) T(W_c) T1(W_f) T1(W_f) T0(W_f)
    (include/bmc_hip.h)."""
    key = (id(wf), id(wc))
    hit = _CHAIN_CACHE.get(key)
    if hit is not None and hit[0]() is wf and hit[1]() is wc and hit[2] == (wf._version, wc._version):
        return hit[3], hit[4]
    dev = wf.device
    s1, s2 = _dense_spec(Cn), _spec2(Cn)
    st = _stream()

    def pack(w, spec, cin):
        out = torch.empty(spec.kpad * Cn, device=dev, dtype=torch.float32)
        lib.call(lib._pack_w, "bmc_pack_weight", w.data_ptr(), spec.kmap(dev).data_ptr(), 1, Cn, cin, 1, spec.kpad, Cn,
                 out.data_ptr(), st)
        return out

    def pack_t(w, spec, cin, src_index):
        out = torch.empty(Cn * Cn, device=dev, dtype=torch.float32)
        lib.call(lib._pack_wt, "bmc_pack_weight_t", w.data_ptr(), spec.kmap(dev).data_ptr(), 1, Cn, cin, 1,
                 src_index * Cn, Cn, Cn, Cn, out.data_ptr(), st)
        return out

    wfd, wcd = wf.detach().contiguous(), wc.detach().contiguous()
    fwd = torch.cat([pack(wfd, s2, 2 * Cn), pack(wcd, s1, Cn)])
    tc, t0, t1 = pack_t(wcd, s1, Cn, 0), pack_t(wfd, s2, 2 * Cn, 0), pack_t(wfd, s2, 2 * Cn, 1)
    bwd = torch.cat([tc, tc, t1, t1, t0])
    if len(_CHAIN_CACHE) > 64:
        for k in [k for k, v in _CHAIN_CACHE.items() if v[0]() is None or v[1]() is None]:
            del _CHAIN_CACHE[k]
    _CHAIN_CACHE[key] = (weakref.ref(wf), weakref.ref(wc), (wf._version, wc._version), fwd, bwd)
    return fwd, bwd


def chain_fwd(s0, s1, wf, bf, gamma, beta, wc, bc, eps, B, H, W, Cn, dev):
    """-> (yhat [B,H,W,C], rstd [B*H*W], centre [B,H,W,C]); s0 / s1: lib.Src of C channels each."""
    fwd, _ = _chain_streams(wf, wc, Cn)
    yhat = torch.empty((B, H, W, Cn), device=dev, dtype=torch.float32)
    centre = torch.empty((B, H, W, Cn), device=dev, dtype=torch.float32)
    rstd = torch.empty(B * H * W, device=dev, dtype=torch.float32)
    a = lib.ChainFwdArgs()
    a.s0, a.s1 = s0, s1
    a.wstream = fwd.data_ptr()
    a.bias_f, a.bias_c, a.gamma, a.beta = bf.data_ptr(), bc.data_ptr(), gamma.data_ptr(), beta.data_ptr()
    a.eps = eps
    a.yhat, a.rstd, a.centre = yhat.data_ptr(), rstd.data_ptr(), centre.data_ptr()
    a.B, a.C, a.H, a.W = B, Cn, H, W
    e0 = ops._prof_begin()
    lib.call(lib._chain_fwd, "bmc_chain_fwd", C.byref(a), _stream())
    ops._prof_end(e0, "chain_kernel<fwd>", 2.0 * B * H * W * Cn * 3 * Cn, B * H * W * (16.0 * Cn + 4))
    return yhat, rstd, centre


def chain_bwd(dc, yhat, rstd, gamma, wf, wc, add, n, H, W, Cn, dev, ds1=None):
    """-> (dz [2n,H,W,C], ds1 [2n,H,W,C], ds0 [n,H,W,C]); dc: lib.Src over 2n launch batches, add: lib.Src or None;
    ds1: optional preallocated destination."""
    _, bwd = _chain_streams(wf, wc, Cn)
    dz = torch.empty((2 * n, H, W, Cn), device=dev, dtype=torch.float32)
    if ds1 is None:
        ds1 = torch.empty((2 * n, H, W, Cn), device=dev, dtype=torch.float32)
    ds0 = torch.empty((n, H, W, Cn), device=dev, dtype=torch.float32)
    a = lib.ChainBwdArgs()
    a.dcentre = dc
    a.wstream = bwd.data_ptr()
    a.gamma, a.yhat, a.rstd = gamma.data_ptr(), yhat.data_ptr(), rstd.data_ptr()
    a.dz, a.ds1, a.ds0 = dz.data_ptr(), ds1.data_ptr(), ds0.data_ptr()
    a.ds0_add = add if add is not None else _null_src()
    a.n, a.C, a.H, a.W = n, Cn, H, W
    e0 = ops._prof_begin()
    lib.call(lib._chain_bwd, "bmc_chain_bwd", C.byref(a), _stream())
    ops._prof_end(e0, "chain_kernel<bwd>", 2.0 * 2 * n * H * W * Cn * 3 * Cn, 2 * n * H * W * (20.0 * Cn + 4))
    return dz, ds1, ds0


def _spec2(c):
    s = _SPEC2.get(c)
    if s is None:
        s = ConvSpec.dense(c, c)
        _SPEC2[c] = s
    return s


def _conv(srcs, w4, spec, owner, bias, out, B, relu=False, residual=None, mask=None, bpg=None, accumulate=False,
          out_b0=0, rule=None):
    """Forward-style launch: out[out_b0 : out_b0+B] = epi(conv(cat(srcs)) + bias)."""
    G, Cout, Cin, taps = w4.shape
    _, H, W, Co = out.shape
    stride = max([x.pix_stride for x in srcs] + [Co] + [x.pix_stride for x in (residual, mask) if x is not None])
    wn = ops.wino_ok(B, H, W, Cout, taps, fwd=not accumulate and mask is None, stride=stride, rule=rule)
    wp = _packed_weight(w4, spec, owner, wino=wn)
    conv_raw(srcs, wp, spec.kpad * taps * coutpad(Cout), bias, Cout if bias is not None else 0,
             out.data_ptr() + 4 * out_b0 * H * W * Co, H * W * Co, Co, B, H, W, Cout, taps, relu=relu, residual=residual,
             bpg=bpg, accumulate=accumulate, mask=mask, flops=2.0 * B * H * W * Cout * taps * spec.kreal, wino=wn)


def _dgrad(g_src, w4, spec, src_index, owner, out, B, residual=None, mask=None, bpg=None, accumulate=False, out_b0=0):
    """Data gradient w.r.t. source `src_index` of the conv with weights w4: out[out_b0:+B] (=|+=) conv^T(g)."""
    G, Cout, Cin, taps = w4.shape
    _, H, W, Co = out.shape
    nch = spec.nch[src_index]
    stride = max([g_src.pix_stride, Co] + [x.pix_stride for x in (residual, mask) if x is not None])
    wn = ops.wino_ok(B, H, W, nch, taps, stride=stride)
    wt = _packed_weight_t(w4, spec, src_index, owner, wino=wn)
    conv_raw([g_src], wt, round_up(Cout, CK) * taps * coutpad(nch), None, 0, out.data_ptr() + 4 * out_b0 * H * W * Co,
             H * W * Co, Co, B, H, W, nch, taps, residual=residual, mask=mask, bpg=bpg, accumulate=accumulate,
             flops=2.0 * B * H * W * spec.real_nch[src_index] * taps * Cout, wino=wn)


def _wgrad(a_src, x_srcs, spec, B, H, W, taps, Cout, dev, w_param, b_param, G=1, w_shape=None, keep=(), window=None):
    # (w_param / b_param: a parameter, None, or for G > 1 a tuple of the G parameters of the weight groups)
    """Weight gradient + bias gradient (column sums of the same A operand, taken from the tiles the pixel-reduction
    GEMM stages anyway) -> (dW, db) for autograd; None where the sums went straight into the leaf parameters' .grad
    (ops.reduce_wgrad).  G > 1 (stacked per-group weights): w_param None, result [G, ...]."""
    # keep: the operand tensors behind a_src / x_srcs (ops.wgrad_side: small launches run on the side stream)
    if ops.wino_wgrad_ok(a_src, x_srcs, spec, taps, Cout, G) and w_param is not None:
        return ops.wgrad_wino(a_src, x_srcs[0], B, H, W, spec, dev, w_param, b_param, w_shape if w_shape is not None else w_param.shape,
                              keep=keep, window=window)
    return ops.wgrad_pgemm(a_src, x_srcs, B, H, W, taps, Cout, spec, dev, w_param, b_param,
                           w_shape if w_shape is not None else w_param.shape, G=G, keep=keep, window=window)


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
        _conv([X(x12)], d(rw1).reshape(1, Cn, Cn, 9), s1, rw1, d(rb1), t12, B2, relu=True, rule=rb1)
        _conv([X(t12)], d(rw2).reshape(1, Cn, Cn, 9), s1, rw2, d(rb2), r12, B2, residual=X(x12), rule=rb2)
        # centres: clustering(LN(convf(cat[xs, other half])))
        fused = chain_supported(Cn)
        if fused:       # one launch; saved for backward: yhat (normalised, before the affine) and rstd
            z12, stats, c12 = chain_fwd(X(xs, mod=n, B=B2), X(x12, shift=n, mod=B2), wf, d(bf), d(gamma), d(beta), wc, d(bc),
                                        eps, B2, H, W, Cn, dev)
            y12 = z12
        else:
            z12, y12, c12 = new(B2), new(B2), new(B2)
            _conv([X(xs, mod=n, B=B2), X(x12, shift=n, mod=B2)], d(wf).reshape(1, Cn, 2 * Cn, 1), s2, wf, d(bf), z12, B2)
            stats = torch.empty(B2 * H * W * 2, device=dev, dtype=torch.float32)
            lib.call(lib._ln_fwd, "bmc_layernorm_fwd", z12.data_ptr(), gamma.data_ptr(), beta.data_ptr(), B2 * H * W, Cn, eps,
                     y12.data_ptr(), stats.data_ptr(), _stream())
            _conv([X(y12)], d(wc).reshape(1, Cn, Cn, 1), s1, wc, d(bc), c12, B2)
        # values: v1 on the first half, v2 on the second (two weight groups)
        wv = ops.stacked((wv1, wv2), lambda: torch.stack([d(wv1).reshape(Cn, Cn, 1), d(wv2).reshape(Cn, Cn, 1)]), "v1x1")
        bv = ops.stacked((bv1, bv2), lambda: torch.stack([d(bv1), d(bv2)]), "stack")
        vfree = vfree_supported(B2 * H * W)
        o12 = new(B2)
        if vfree:       # attention without v (above): att = scale (G0 W_v^T + s b_v^T), out = (P W_v) x + P b_v
            G0, sc = _gram_on_x(X(c12), X(x12), B2, H, W, Cn, dev)
            wg, v12 = wv.view(2, Cn, Cn), None
            att = _times_wt(G0, sc, wg, bv, n, scale, torch.empty_like(G0))
            p = torch.empty_like(att)
            lib.call(lib._sm_fwd, "bmc_softmax_fwd", att.data_ptr(), B2 * Cn, Cn, p.data_ptr(), _stream())
            pw, pb = torch.empty_like(p), torch.empty_like(sc)                                 # P W_v [b, i, k],  P b_v [b, i]
            _times_w(p, wg, bv, n, pw, (Cn * Cn, Cn, 1), vec=pb)
            _conv([X(x12)], pw.view(B2, Cn, Cn, 1), s1, None, pb, o12, B2, residual=X(r12, shift=n, mod=B2), bpg=1)
        else:
            G0 = sc = None
            v12 = new(B2)
            _conv([X(x12)], wv, s1, wv, bv, v12, B2, bpg=n)
            # channel attention per sample
            slabs, nsplit, G = pgemm_raw(X(c12), [X(v12)], B2, H, W, 1, 1, Cn, Cn, dev, flops=2.0 * B2 * H * W * Cn * Cn)
            att = torch.empty((B2, Cn, Cn), device=dev, dtype=torch.float32)
            lib.call(lib._red_p, "bmc_pgemm_reduce_plain", slabs.data_ptr(), nsplit, G, Cn, Cn, scale, att.data_ptr(), _stream())
            p = torch.empty_like(att)
            lib.call(lib._sm_fwd, "bmc_softmax_fwd", att.data_ptr(), B2 * Cn, Cn, p.data_ptr(), _stream())
            _conv([X(v12)], p.view(B2, Cn, Cn, 1), s1, None, None, o12, B2, residual=X(r12, shift=n, mod=B2), bpg=1)
        # shared stream: unclustering(cat[c1, c2]) + xs
        xs_new = new(n)
        _conv([X(c12, b0=0, B=n), X(c12, b0=n, B=n)], d(wu).reshape(1, Cn, 2 * Cn, 1), s2, wu, d(bu), xs_new, n, residual=X(xs))
        ctx.save_for_backward(x12, xs, t12, z12, stats, y12, c12, v12 if not vfree else G0, p, rw1, rw2, wf, gamma, wc, wu, wv1, wv2, beta,
                              *((sc, bv1, bv2) if vfree else ()))
        ctx.vfree = vfree
        ctx.owners = (rw1, rw2, wf, wc, wu)
        ctx.window = ops.current_window()
        ctx.params = (rw1, rb1, rw2, rb2, wf, bf, gamma, beta, wc, bc, wu, bu)     # the caller's objects (gradient sinks)
        ctx.vparams = (wv1, wv2, bv1, bv2)
        ctx.scale = scale
        ctx.fused = fused
        ctx.gslot = getattr(x12, "_bmc_gslot", None)     # (set before .contiguous(): see the top of this function)
        return o12, xs_new

    @staticmethod
    def backward(ctx, do12, dxs_new):
        saved = ctx.saved_tensors           # (once: a second access breaks torch.utils.checkpoint's unpack bookkeeping)
        x12, xs, t12, z12, stats, y12, c12, v12, p, rw1, rw2, wf, gamma, wc, wu, wv1, wv2, beta = saved[:18]
        vfree = ctx.vfree
        if vfree:
            G0, (sc, bv1, bv2) = v12, saved[18:]
        o_rw1, o_rw2, o_wf, o_wc, o_wu = ctx.owners
        p_rw1, p_rb1, p_rw2, p_rb2, p_wf, p_bf, p_gamma, p_beta, p_wc, p_bc, p_wu, p_bu = ctx.params
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
        # (keyed by the caller's parameter objects: the saved tensors are fresh aliases in every backward)
        w_v = ops.stacked(ctx.vparams[:2], lambda: torch.stack([wv1.detach().reshape(Cn, Cn, 1), wv2.detach().reshape(Cn, Cn, 1)]), "v1x1")

        # ---- out = P v (+ rotated residual): dP, dv
        if vfree:       # dM = g_o^T x, t = column sums of g_o;  dP = dM W_v^T + t b_v^T
            wg = w_v.view(2, Cn, Cn)
            bg = ops.stacked(ctx.vparams[2:], lambda: torch.stack([bv1.detach(), bv2.detach()]), "stack")
            dM, tg = _gram_on_x(X(g_o), X(x12), B2, H, W, Cn, dev)
            dp = _times_wt(dM, tg, wg, bg, n, 1.0, torch.empty_like(dM))
        else:
            slabs, nsplit, G = pgemm_raw(X(g_o), [X(v12)], B2, H, W, 1, 1, Cn, Cn, dev, flops=2.0 * B2 * H * W * Cn * Cn)
            dp = torch.empty_like(p)
            lib.call(lib._red_p, "bmc_pgemm_reduce_plain", slabs.data_ptr(), nsplit, G, Cn, Cn, 1.0, dp.data_ptr(), _stream())
        # ---- softmax, Gram (att = scale * center v^T)
        da = torch.empty_like(p)
        lib.call(lib._sm_bwd, "bmc_softmax_bwd", p.data_ptr(), dp.data_ptr(), B2 * Cn, Cn, ctx.scale, da.data_ptr(), _stream())
        if vfree:
            # value-convolution parameters: dW_v = P^T dM + da^T G0, db_v = P^T t + da^T s (summed over the group's samples)
            dwv, dbv = _value_param_grads(p, dM, tg, da, G0, sc, ctx.vparams[:2], ctx.vparams[2:], B2 * H * W)
        # Each of dv and d center has two contributions that are 1x1 products with per-sample / per-half matrices: ONE
        # two-source launch each (K = 2C, the matrices side by side) instead of a launch + accumulating launches --
        #   dv[b]      = P_b^T g_o[b] + da_b^T center[b]
        #   dcenter[b] = da_b  v[b]   + W_u[:, half(b)]^T g_x[b mod n]            (unclustering reads cat[c1, c2])
        dc12 = new(B2)
        # [B2, C (c_i), C (co)], cached per version of the unclustering weight and batch size (it is the same in all 8 windows)
        w_ut = ops.stacked((p_wu,), lambda: wu.detach().view(Cn, 2, Cn).permute(1, 2, 0).repeat_interleave(n, 0).contiguous(), "wut%d" % n)
        if vfree:
            # without v:  dx12 (+)= (P W_v)^T g_o + (da W_v)^T center  (below, once dx12 exists);
            #             dcenter = (da W_v) x + da b_v + W_u[:, half]^T g_x
            # the per-sample matrices are written straight into the two-source weight layouts [b][cout][cin of source 0 | 1]
            w_dx = torch.empty((B2, Cn, 2 * Cn, 1), device=dev, dtype=torch.float32)
            w_dc = torch.empty((B2, Cn, 2 * Cn, 1), device=dev, dtype=torch.float32)
            w_dc.view(B2, Cn, 2 * Cn)[:, :, Cn:] = w_ut
            ds = torch.empty((B2, Cn), device=dev, dtype=torch.float32)
            cc2 = 2 * Cn * Cn
            _times_w(da, wg, bg, n, w_dc, (cc2, 2 * Cn, 1), vec=ds)                            # [da W_v | .],  da b_v
            _times_w(da, wg, bg, n, w_dx[:, :, Cn:], (cc2, 1, 2 * Cn))                         # [. | (da W_v)^T]
            _times_w(p, wg, bg, n, w_dx, (cc2, 1, 2 * Cn))                                     # [(P W_v)^T | .]
            _conv([X(x12), X(g_x, mod=n, B=B2)], w_dc, s2, None, ds, dc12, B2, bpg=1)
        else:
            dv12 = new(B2)
            w_dv = torch.cat([p.transpose(1, 2), da.transpose(1, 2)], 2).view(B2, Cn, 2 * Cn, 1)
            _conv([X(g_o), X(c12)], w_dv, s2, None, None, dv12, B2, bpg=1)
            w_dc = torch.cat([da, w_ut], 2).view(B2, Cn, 2 * Cn, 1)
            _conv([X(v12), X(g_x, mod=n, B=B2)], w_dc, s2, None, None, dc12, B2, bpg=1)
        # ---- unclustering(cat[c1, c2]) + xs: weight gradient
        dwu, dbu = _wgrad(X(g_x), [X(c12, b0=0, B=n), X(c12, b0=n, B=n)], s2, n, H, W, 1, Cn, dev, p_wu, p_bu, keep=(g_x, c12), window=ctx.window)
        # ---- value convs (two weight groups)
        if not vfree:
            dwv, dbv = _wgrad(X(dv12), [X(x12)], s1, B2, H, W, 1, Cn, dev, ctx.vparams[:2], ctx.vparams[2:], G=2,
                              w_shape=(2, Cn, Cn, 1, 1), keep=(dv12, x12), window=ctx.window)
        if ctx.fused:
            # ---- clustering, LayerNorm, convf: ONE data-gradient launch (csrc/chain.hip); y12 holds yhat, stats rstd.
            # dx12 = conv_f^T (second half of its inputs), dxs = skip + conv_f^T (first half) summed over both halves.
            dz12, dx12, dxs = chain_bwd(X(dc12), y12, stats, gamma.detach(), wf, wc, X(g_x), n, H, W, Cn, dev,
                                        ds1=ops.grad_slot(ctx.gslot, x12))
            # G = dc^T yhat and the clustering bias gradient (temporaries: dW_c, dgamma, dbeta follow from them)
            Gm, dbc_t = _wgrad(X(dc12), [X(y12)], s1, B2, H, W, 1, Cn, dev, None, None, w_shape=(Cn, Cn))
            sg = ops.sink_group([p_wc, p_bc, p_gamma, p_beta])
            if sg is not None:
                (o_w, o_b, o_g, o_bt), acc = sg
                dwc = dbc = dgamma = dbeta = None
            else:
                o_w, o_b, acc = Gm, None, 0
                o_g = dgamma = torch.empty(Cn, device=dev, dtype=torch.float32)
                o_bt = dbeta = torch.empty(Cn, device=dev, dtype=torch.float32)
                dwc, dbc = Gm, dbc_t
            lib.call(lib._chain_affine, "bmc_chain_affine_grads", Gm.data_ptr(), dbc_t.data_ptr(), wc.detach().data_ptr(),
                     gamma.detach().data_ptr(), beta.detach().data_ptr(), Cn, o_w.data_ptr(),
                     o_b.data_ptr() if o_b is not None else None, o_g.data_ptr(), o_bt.data_ptr(), acc, _stream())
            dwf, dbf = _wgrad(X(dz12), [X(xs, mod=n, B=B2), X(x12, shift=n, mod=B2)], s2, B2, H, W, 1, Cn, dev, p_wf, p_bf,
                              keep=(dz12, xs, x12), window=ctx.window)
            if vfree:
                _conv([X(g_o), X(c12)], w_dx, s2, None, None, dx12, B2, bpg=1, accumulate=True)         # dx12 += attention
            else:
                _dgrad(X(dv12), w_v, s1, 0, w_v, dx12, B2, bpg=n, accumulate=True)                      # dx12 += value convs
        else:
            dx12 = ops.grad_slot(ctx.gslot, x12)
            if vfree:
                _conv([X(g_o), X(c12)], w_dx, s2, None, None, dx12, B2, bpg=1)
            else:
                _dgrad(X(dv12), w_v, s1, 0, w_v, dx12, B2, bpg=n)                                       # dx12  =
            # ---- clustering, LayerNorm, convf
            dwc, dbc = _wgrad(X(dc12), [X(y12)], s1, B2, H, W, 1, Cn, dev, p_wc, p_bc, keep=(dc12, y12), window=ctx.window)
            dy12 = new(B2)
            _dgrad(X(dc12), w_c, s1, 0, o_wc, dy12, B2)
            dz12 = new(B2)
            ws = torch.empty(2 * 1024 * Cn, device=dev, dtype=torch.float32)
            sg = ops.sink_group([p_gamma, p_beta])
            if sg is not None:
                (o_g, o_bt), ln_acc = sg
                dgamma = dbeta = None
            else:
                o_g = dgamma = torch.empty(Cn, device=dev, dtype=torch.float32)
                o_bt = dbeta = torch.empty(Cn, device=dev, dtype=torch.float32)
                ln_acc = 0
            lib.call(lib._ln_bwd, "bmc_layernorm_bwd", dy12.data_ptr(), z12.data_ptr(), stats.data_ptr(), gamma.data_ptr(),
                     B2 * H * W, Cn, dz12.data_ptr(), ws.data_ptr(), o_g.data_ptr(), o_bt.data_ptr(), ln_acc, _stream())
            dwf, dbf = _wgrad(X(dz12), [X(xs, mod=n, B=B2), X(x12, shift=n, mod=B2)], s2, B2, H, W, 1, Cn, dev, p_wf, p_bf,
                              keep=(dz12, xs, x12), window=ctx.window)
            dxs = new(n)
            _dgrad(X(dz12, b0=0, B=n), w_f, s2, 0, o_wf, dxs, n, residual=X(g_x))                        # dxs  = skip + half 0
            _dgrad(X(dz12, b0=n, B=n), w_f, s2, 0, o_wf, dxs, n, accumulate=True)                        # dxs += half 1
            _dgrad(X(dz12, shift=n, mod=B2), w_f, s2, 1, o_wf, dx12, B2, accumulate=True)                # dx12 += (rotated)
        # ---- residual block, upstream gradient = batch-rotated g_o
        g_r = X(g_o, shift=n, mod=B2)
        dw2, db2 = _wgrad(g_r, [X(t12)], s1, B2, H, W, 9, Cn, dev, p_rw2, p_rb2, keep=(g_o, t12), window=ctx.window)
        dt = new(B2)
        _dgrad(g_r, w_r2, s1, 0, o_rw2, dt, B2, mask=X(t12))
        dw1, db1 = _wgrad(X(dt), [X(x12)], s1, B2, H, W, 9, Cn, dev, p_rw1, p_rb1, keep=(dt, x12), window=ctx.window)
        _dgrad(X(dt), w_r1, s1, 0, o_rw1, dx12, B2, residual=g_r, accumulate=True)                       # dx12 += conv1^T + skip
        v = lambda t, ref: None if t is None else t.view(ref.shape)
        gv = (None,) * 4 if dwv is None else (dwv[0].reshape(wv1.shape), dbv[0], dwv[1].reshape(wv2.shape), dbv[1])
        return (dx12, dxs, dw1, db1, dw2, db2, dwf, dbf, dgamma, dbeta, v(dwc, wc), dbc, dwu, dbu, *gv, None, None)


class BIEFirstFn(torch.autograd.Function):
    """BIETwinFn without everything that only feeds its SECOND output (the last ParallelBlk of the backbone never reads it,
    models/BMCNet.py:75-82): x12 [2n,H,W,C] = [first; second], xs [n,H,W,C] ->
        o1 = softmax(att1) v1(first) + Res(second)   [n],   xs_new = unclustering(cat[c1, c2]) + xs   [n].
    The residual block runs on the second half only, values / Gram / softmax / attention on the first half only; both
    centres are still needed (unclustering) and come from the one fused chain launch.  Needs the fused chain."""

    @staticmethod
    def forward(ctx, x12, xs, rw1, rb1, rw2, rb2, wf, bf, gamma, beta, wc, bc, wu, bu, wv1, bv1, scale, eps):
        _need_gpu(x12)
        x12, xs = x12.contiguous(), xs.contiguous()
        B2, H, W, Cn = x12.shape
        n = B2 // 2
        dev = x12.device
        s1, s2 = _dense_spec(Cn), _spec2(Cn)
        new = lambda b: torch.empty((b, H, W, Cn), device=dev, dtype=torch.float32)
        X = lambda t, **k: _src(t, 0, Cn, k.get("shift", 0), k.get("mod"), k.get("b0", 0), k.get("B", t.shape[0]))
        d = lambda t: t.detach()
        second = X(x12, b0=n, B=n)
        t2, r2 = new(n), new(n)
        _conv([second], d(rw1).reshape(1, Cn, Cn, 9), s1, rw1, d(rb1), t2, n, relu=True, rule=rb1)
        _conv([X(t2)], d(rw2).reshape(1, Cn, Cn, 9), s1, rw2, d(rb2), r2, n, residual=second, rule=rb2)
        yhat, rstd, c12 = chain_fwd(X(xs, mod=n, B=B2), X(x12, shift=n, mod=B2), wf, d(bf), d(gamma), d(beta), wc, d(bc), eps,
                                    B2, H, W, Cn, dev)
        vfree = vfree_supported(B2 * H * W)
        o1 = new(n)
        if vfree:       # attention without v (top of this file)
            G0, sc = _gram_on_x(X(c12, b0=0, B=n), X(x12, b0=0, B=n), n, H, W, Cn, dev)
            wg, bg, v1 = d(wv1).reshape(1, Cn, Cn), d(bv1).reshape(1, Cn), None
            att = _times_wt(G0, sc, wg, bg, n, scale, torch.empty_like(G0))
            p = torch.empty_like(att)
            lib.call(lib._sm_fwd, "bmc_softmax_fwd", att.data_ptr(), n * Cn, Cn, p.data_ptr(), _stream())
            pw, pb = torch.empty_like(p), torch.empty_like(sc)
            _times_w(p, wg, bg, n, pw, (Cn * Cn, Cn, 1), vec=pb)
            _conv([X(x12, b0=0, B=n)], pw.view(n, Cn, Cn, 1), s1, None, pb, o1, n, residual=X(r2), bpg=1)
        else:
            G0 = sc = None
            v1 = new(n)
            _conv([X(x12, b0=0, B=n)], d(wv1).reshape(1, Cn, Cn, 1), s1, wv1, d(bv1), v1, n)
            slabs, nsplit, G = pgemm_raw(X(c12, b0=0, B=n), [X(v1)], n, H, W, 1, 1, Cn, Cn, dev, flops=2.0 * n * H * W * Cn * Cn)
            att = torch.empty((n, Cn, Cn), device=dev, dtype=torch.float32)
            lib.call(lib._red_p, "bmc_pgemm_reduce_plain", slabs.data_ptr(), nsplit, G, Cn, Cn, scale, att.data_ptr(), _stream())
            p = torch.empty_like(att)
            lib.call(lib._sm_fwd, "bmc_softmax_fwd", att.data_ptr(), n * Cn, Cn, p.data_ptr(), _stream())
            _conv([X(v1)], p.view(n, Cn, Cn, 1), s1, None, None, o1, n, residual=X(r2), bpg=1)
        xs_new = new(n)
        _conv([X(c12, b0=0, B=n), X(c12, b0=n, B=n)], d(wu).reshape(1, Cn, 2 * Cn, 1), s2, wu, d(bu), xs_new, n, residual=X(xs))
        ctx.save_for_backward(x12, xs, t2, yhat, rstd, c12, v1 if not vfree else G0, p, rw1, rw2, wf, gamma, wc, wu, wv1, beta,
                              *((sc, bv1) if vfree else ()))
        ctx.vfree = vfree
        ctx.owners = (rw1, rw2, wf, wc, wu, wv1)
        ctx.window = ops.current_window()
        ctx.params = (rw1, rb1, rw2, rb2, wf, bf, gamma, beta, wc, bc, wu, bu, wv1, bv1)     # the caller's objects (gradient sinks)
        ctx.scale = scale
        return o1, xs_new

    @staticmethod
    def backward(ctx, do1, dxs_new):
        saved = ctx.saved_tensors
        x12, xs, t2, yhat, rstd, c12, v1, p, rw1, rw2, wf, gamma, wc, wu, wv1, beta = saved[:16]
        vfree = ctx.vfree
        if vfree:
            G0, (sc, bv1) = v1, saved[16:]
        o_rw1, o_rw2, o_wf, o_wc, o_wu, o_wv1 = ctx.owners
        p_rw1, p_rb1, p_rw2, p_rb2, p_wf, p_bf, p_gamma, p_beta, p_wc, p_bc, p_wu, p_bu, p_wv1, p_bv1 = ctx.params
        B2, H, W, Cn = x12.shape
        n = B2 // 2
        dev = x12.device
        s1, s2 = _dense_spec(Cn), _spec2(Cn)
        new = lambda b: torch.empty((b, H, W, Cn), device=dev, dtype=torch.float32)
        X = lambda t, **k: _src(t, 0, Cn, k.get("shift", 0), k.get("mod"), k.get("b0", 0), k.get("B", t.shape[0]))
        g_o = do1.contiguous() if do1 is not None else torch.zeros_like(t2)
        g_x = dxs_new.contiguous() if dxs_new is not None else torch.zeros_like(xs)
        w_r1, w_r2 = rw1.detach().reshape(1, Cn, Cn, 9), rw2.detach().reshape(1, Cn, Cn, 9)
        w_v1 = wv1.detach().reshape(1, Cn, Cn, 1)
        # ---- o1 = P v1 + r2: dP, softmax, Gram
        if vfree:
            wg, bg = wv1.detach().reshape(1, Cn, Cn), bv1.detach().reshape(1, Cn)
            dM, tg = _gram_on_x(X(g_o), X(x12, b0=0, B=n), n, H, W, Cn, dev)
            dp = _times_wt(dM, tg, wg, bg, n, 1.0, torch.empty_like(dM))
        else:
            slabs, nsplit, G = pgemm_raw(X(g_o), [X(v1)], n, H, W, 1, 1, Cn, Cn, dev, flops=2.0 * n * H * W * Cn * Cn)
            dp = torch.empty_like(p)
            lib.call(lib._red_p, "bmc_pgemm_reduce_plain", slabs.data_ptr(), nsplit, G, Cn, Cn, 1.0, dp.data_ptr(), _stream())
        da = torch.empty_like(p)
        lib.call(lib._sm_bwd, "bmc_softmax_bwd", p.data_ptr(), dp.data_ptr(), n * Cn, Cn, ctx.scale, da.data_ptr(), _stream())
        if vfree:
            dwv1, dbv1 = _value_param_grads(p, dM, tg, da, G0, sc, (p_wv1,), (p_bv1,), n * H * W)
            if dwv1 is not None:
                dwv1, dbv1 = dwv1.view(wv1.shape), dbv1.view(Cn)
        #   dv1[b]     = P_b^T g_o[b] + da_b^T c1[b]
        #   dcentre[b] = (b < n: da_b v1[b]) + W_u[:, half(b)]^T g_x[b mod n]     (the second half has no attention term: a zero matrix)
        dc12 = new(B2)
        # [B2, C (c_i), C (co)], cached per version of the unclustering weight and batch size (it is the same in all 8 windows)
        w_ut = ops.stacked((p_wu,), lambda: wu.detach().view(Cn, 2, Cn).permute(1, 2, 0).repeat_interleave(n, 0).contiguous(), "wut%d" % n)
        if vfree:       # (as in BIETwinFn; the second half has no attention term: zero matrix, zero bias)
            w_dx = torch.empty((n, Cn, 2 * Cn, 1), device=dev, dtype=torch.float32)
            w_dc = torch.zeros((B2, Cn, 2 * Cn, 1), device=dev, dtype=torch.float32)
            w_dc.view(B2, Cn, 2 * Cn)[:, :, Cn:] = w_ut
            ds = torch.zeros((B2, Cn), device=dev, dtype=torch.float32)
            cc2 = 2 * Cn * Cn
            _times_w(da, wg, bg, n, w_dc, (cc2, 2 * Cn, 1), vec=ds)                            # (first n samples; the rest stay zero)
            _times_w(da, wg, bg, n, w_dx[:, :, Cn:], (cc2, 1, 2 * Cn))
            _times_w(p, wg, bg, n, w_dx, (cc2, 1, 2 * Cn))
            _conv([X(x12), X(g_x, mod=n, B=B2)], w_dc, s2, None, ds, dc12, B2, bpg=1)
        else:
            dv1 = new(n)
            w_dv = torch.cat([p.transpose(1, 2), da.transpose(1, 2)], 2).view(n, Cn, 2 * Cn, 1)
            _conv([X(g_o), X(c12, b0=0, B=n)], w_dv, s2, None, None, dv1, n, bpg=1)
            w_dc = torch.cat([torch.cat([da, torch.zeros_like(da)], 0), w_ut], 2).view(B2, Cn, 2 * Cn, 1)
            _conv([X(v1, mod=n, B=B2), X(g_x, mod=n, B=B2)], w_dc, s2, None, None, dc12, B2, bpg=1)
        dwu, dbu = _wgrad(X(g_x), [X(c12, b0=0, B=n), X(c12, b0=n, B=n)], s2, n, H, W, 1, Cn, dev, p_wu, p_bu, keep=(g_x, c12), window=ctx.window)
        if not vfree:
            dwv1, dbv1 = _wgrad(X(dv1), [X(x12, b0=0, B=n)], s1, n, H, W, 1, Cn, dev, p_wv1, p_bv1, keep=(dv1, x12), window=ctx.window)
        # ---- clustering, LayerNorm, convf (csrc/chain.hip), as in BIETwinFn
        dz12, dx12, dxs = chain_bwd(X(dc12), yhat, rstd, gamma.detach(), wf, wc, X(g_x), n, H, W, Cn, dev)
        Gm, dbc_t = _wgrad(X(dc12), [X(yhat)], s1, B2, H, W, 1, Cn, dev, None, None, w_shape=(Cn, Cn))
        sg = ops.sink_group([p_wc, p_bc, p_gamma, p_beta])
        if sg is not None:
            (o_w, o_b, o_g, o_bt), acc = sg
            dwc = dbc = dgamma = dbeta = None
        else:
            o_w, o_b, acc = Gm, None, 0
            o_g = dgamma = torch.empty(Cn, device=dev, dtype=torch.float32)
            o_bt = dbeta = torch.empty(Cn, device=dev, dtype=torch.float32)
            dwc, dbc = Gm, dbc_t
        lib.call(lib._chain_affine, "bmc_chain_affine_grads", Gm.data_ptr(), dbc_t.data_ptr(), wc.detach().data_ptr(),
                 gamma.detach().data_ptr(), beta.detach().data_ptr(), Cn, o_w.data_ptr(),
                 o_b.data_ptr() if o_b is not None else None, o_g.data_ptr(), o_bt.data_ptr(), acc, _stream())
        dwf, dbf = _wgrad(X(dz12), [X(xs, mod=n, B=B2), X(x12, shift=n, mod=B2)], s2, B2, H, W, 1, Cn, dev, p_wf, p_bf,
                          keep=(dz12, xs, x12), window=ctx.window)
        if vfree:
            _conv([X(g_o), X(c12, b0=0, B=n)], w_dx, s2, None, None, dx12, n, bpg=1, accumulate=True)    # dx12[first] += attention
        else:
            _dgrad(X(dv1), w_v1, s1, 0, o_wv1, dx12, n, accumulate=True)                                 # dx12[first] += value conv
        # ---- residual block (second half), upstream gradient = g_o
        dw2, db2 = _wgrad(X(g_o), [X(t2)], s1, n, H, W, 9, Cn, dev, p_rw2, p_rb2, keep=(g_o, t2), window=ctx.window)
        dt = new(n)
        _dgrad(X(g_o), w_r2, s1, 0, o_rw2, dt, n, mask=X(t2))
        dw1, db1 = _wgrad(X(dt), [X(x12, b0=n, B=n)], s1, n, H, W, 9, Cn, dev, p_rw1, p_rb1, keep=(dt, x12), window=ctx.window)
        _dgrad(X(dt), w_r1, s1, 0, o_rw1, dx12, n, residual=X(g_o), accumulate=True, out_b0=n)           # dx12[second] += conv1^T + skip
        v = lambda t, ref: None if t is None else t.view(ref.shape)
        return (dx12, dxs, dw1, db1, dw2, db2, dwf, dbf, dgamma, dbeta, v(dwc, wc), dbc, dwu, dbu, v(dwv1, wv1), dbv1, None, None)


def bie_first(m, x12, xs):
    """m: models.submodules.BIE; -> (o1, xs_new) (BIEFirstFn)."""
    r = m.conv1
    return BIEFirstFn.apply(x12, xs, r.conv1.weight, r.conv1.bias, r.conv2.weight, r.conv2.bias, m.convf1.weight,
                            m.convf1.bias, m.norm_s.weight, m.norm_s.bias, m.clustering.weight, m.clustering.bias,
                            m.unclustering.weight, m.unclustering.bias, m.v1.weight, m.v1.bias, m.scale, m.norm_s.eps)


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
    def forward(ctx, t, n=None):
        # (n: split point, default the middle -- models/BMCNet.py's [xs_p_st; xs_n_st | xs] is a 2B / B split)
        n = t.shape[0] // 2 if n is None else n
        ctx.n = n
        ctx.shape = t.shape
        return t[:n], t[n:]

    @staticmethod
    def unstack(t):
        """apply() + the gradient-pair tags on the two views (ops.GradPair): consumers that know the protocol write their
        input gradients into the halves of one buffer, and backward below hands it on without a copy."""
        a, b = Unstack2Fn.apply(t)
        if t.requires_grad and a.grad_fn is not None:
            pair = ops.GradPair(t.shape[0] // 2, t.shape)
            a._bmc_gslot, b._bmc_gslot = (pair, 0), (pair, 1)
            a.grad_fn.pair = pair        # backward drops the pair's reference to the buffer (graph nodes outlive their backward)
        return a, b

    @staticmethod
    def backward(ctx, g1, g2):
        if g1 is not None and g2 is not None and g1.is_contiguous() and g2.is_contiguous() \
                and g1.untyped_storage().data_ptr() == g2.untyped_storage().data_ptr() \
                and g2.storage_offset() == g1.storage_offset() + g1.numel():
            # the two gradients already lie back to back in one buffer (ops.StackViewsFn.backward hands out such views)
            out = torch.empty(0, device=g1.device, dtype=g1.dtype)
            out.set_(g1.untyped_storage(), g1.storage_offset(), tuple(ctx.shape), torch.empty(ctx.shape, device="meta").stride())
            pair = getattr(ctx, "pair", None)
            if pair is not None:
                pair.buf = None          # `out` owns the storage from here on
            return out, None
        pair = getattr(ctx, "pair", None)
        if pair is not None:
            pair.buf = None
        z = lambda k: torch.zeros((k,) + tuple(ctx.shape[1:]), device=(g1 if g1 is not None else g2).device)
        return torch.cat([g1 if g1 is not None else z(ctx.n), g2 if g2 is not None else z(ctx.shape[0] - ctx.n)], 0), None
