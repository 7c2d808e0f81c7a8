"""torch.autograd.Function wrappers around the libbmc_hip.so kernels.

Everything here works on fp32 NHWC tensors ([B,H,W,C], contiguous) living on
an MI355X; there is no CPU path (a CPU tensor raises).  The reference computes
the same quantities with ATen ops + autograd (models/submodules.py,
models/BMCNet.py); each function cites what it stands in for.
"""
from __future__ import annotations

import ctypes as C
import os
import threading
import weakref
from typing import List, Optional, Sequence

import torch

from . import lib

CK = 16
_cur_dev = torch._C._cuda_getDevice      # torch.cuda.current_device() without its Python-level lazy-init wrapper


_STREAM_OVERRIDE = None      # raw handle of the side stream while a weight-gradient launch is being issued there (wgrad_side),
_STREAM_OVERRIDE_TS = None   # the same stream as a torch.cuda.Stream


def _stream():
    # torch.cuda.current_stream().cuda_stream without the Python-level wrappers (9 us -> 0.3 us per launch: the step is
    # host-bound at small frame sizes)
    if _STREAM_OVERRIDE is not None:
        return _STREAM_OVERRIDE
    return torch._C._cuda_getCurrentRawStream(_cur_dev())


def _on_side(*tensors):
    """Workspaces torch allocated (on the launch stream's pool) for kernels that run on the side stream: the allocator must
    not hand their memory out again before the side stream is done with it."""
    if _STREAM_OVERRIDE is not None:
        for t in tensors:
            if t is not None:
                t.record_stream(_STREAM_OVERRIDE_TS)


def _need_gpu(t: torch.Tensor):
    if not t.is_cuda:
        raise RuntimeError("bmc_hip: tensor is on %s -- the BMCNet MI355X path has no CPU fallback" % t.device)
    if t.dtype != torch.float32:
        raise RuntimeError("bmc_hip: fp32 tensors only (got %s)" % t.dtype)
    if t.device.index != _cur_dev():
        # launches go to the CURRENT device's current stream (_stream()): a tensor of another device would be
        # touched from the wrong device / stream
        raise RuntimeError("bmc_hip: tensor lives on %s but the current device is cuda:%d -- call torch.cuda.set_device "
                           "(one process per GPU)" % (t.device, _cur_dev()))


def round_up(v, m):
    return (v + m - 1) // m * m


def coutpad(c):
    return 32 if c <= 32 else round_up(c, 128)


class View:
    """Channel/batch window of an NHWC tensor used as a convolution operand.

    Launch batch b reads batch ((b + shift) % mod) + b0 of `t`, channels
    [c0, c0 + nch)."""
    __slots__ = ("t", "c0", "nch", "shift", "mod", "b0")

    def __init__(self, t, c0=0, nch=None, shift=0, mod=None, b0=0):
        self.t = t
        self.c0 = c0
        self.nch = t.shape[3] - c0 if nch is None else nch
        self.shift = shift
        self.mod = mod
        self.b0 = b0

    def meta(self):
        return (self.c0, self.nch, self.shift, self.mod, self.b0)


def _src(t: torch.Tensor, c0, nch, shift, mod, b0, launch_b) -> lib.Src:
    Bt, H, W, Ct = t.shape
    s = lib.Src()
    s.ptr = t.data_ptr() + 4 * (b0 * H * W * Ct + c0)
    s.batch_stride = H * W * Ct
    s.pix_stride = Ct
    s.nch = nch
    s.batch_shift = shift
    s.batch_mod = mod if mod is not None else max(launch_b, 1) + shift
    return s


def _null_src() -> lib.Src:
    s = lib.Src()
    s.ptr = None
    s.batch_mod = 1
    return s


# --------------------------------------------------------------------------
# conv spec: how the reference's concatenated input channels map onto packed,
# 16-channel-granular sources
# --------------------------------------------------------------------------
class ConvSpec:
    """sources: list of per-source lists giving, for each physical channel of the
    source window, the reference input-channel index it carries (or -1 = padding)."""

    def __init__(self, sources: Sequence[Sequence[int]], cin: Optional[int] = None):
        """cin: input channels of the weight tensor the launch is given (default: just enough for the channels the sources
        name).  A spec may name only PART of them -- one launch of a convolution that is evaluated as several launches over
        channel subsets of ONE parameter (Backbone.conv_fs): packing reads, and the weight gradient writes, only the named
        columns of the full [Cout, cin, kh, kw] tensor."""
        self.nch = [len(s) for s in sources]
        for n in self.nch:
            assert n % CK == 0, "source channel windows must be multiples of 16"
        flat = [c for s in sources for c in s]
        self.kpad = len(flat)
        self.kmap_host = flat
        self.cin = max(flat) + 1 if cin is None else cin
        assert self.cin > max(flat)
        self.covers_all = set(range(self.cin)) <= set(flat)     # does a weight gradient through this spec define every column?
        self.kreal = sum(1 for c in flat if c >= 0)             # input channels the launch really contracts (algorithmic FLOPs)
        self.real_nch = [sum(1 for c in src if c >= 0) for src in sources]
        self._kmap = {}
        self._packs = {}

    def kmap(self, device):
        k = self._kmap.get(device)
        if k is None:
            k = torch.tensor(self.kmap_host, dtype=torch.int32, device=device)
            self._kmap[device] = k
        return k

    @staticmethod
    def dense(*nchs):
        """Sources carry the reference's channels in order, no padding."""
        out, c = [], 0
        for n in nchs:
            out.append(list(range(c, c + n)))
            c += n
        return ConvSpec(out)



# Optional in-situ kernel timing (bench.py): when PROFILE is a list, every conv / pgemm launch appends
# (kernel kind, algorithmic FLOPs, start event, end event, algorithmic HBM bytes) recorded on the launch stream.  The bytes are
# every operand read once and the output written once (DESIGN.md's per-pixel figures x the launch's pixels).
PROFILE = None
PROFILE_WINO = [0, 0]       # while PROFILE is a list: conv launches that took the Winograd kernel / the direct kernels


def _prof_record(e):
    # on the stream the launch really goes to (torch's current stream, or the side stream of wgrad_side)
    if _STREAM_OVERRIDE is not None:
        e.record(_STREAM_OVERRIDE_TS)
    else:
        e.record()


def _prof_begin():
    if PROFILE is None:
        return None
    e = torch.cuda.Event(enable_timing=True)
    _prof_record(e)
    return e


def _prof_end(e0, kind, flops, nbytes=0.0):
    if e0 is not None:
        e1 = torch.cuda.Event(enable_timing=True)
        _prof_record(e1)
        PROFILE.append((kind, flops, e0, e1, nbytes))


# Arithmetic of the convolution kernels (include/bmc_hip.h BMC_MATH_*): 0 = native fp32 MFMA (default, the headline
# path), 1 = bf16 operands / fp32 accumulate, 3 = fp32 operands split into three bf16 planes, six plane products
# (fp32-equivalent).  Set with set_math() or BMC_MATH=fp32|bf16|bf16x6 before the first launch.
MATH_NAMES = {"fp32": 0, "bf16": 1, "bf16x6": 3}
MATH = MATH_NAMES[os.environ.get("BMC_MATH", "fp32")]


def set_math(mode):
    """mode: "fp32" | "bf16" | "bf16x6" (or 0 / 1 / 3).  Packed-weight caches are keyed by the mode."""
    global MATH
    MATH = MATH_NAMES[mode] if isinstance(mode, str) else int(mode)
    if MATH not in (0, 1, 3):
        raise ValueError("unknown math mode %r" % (mode,))


def _split_planes(packed: torch.Tensor, cp: int):
    """fp32 pack [nsteps][cp][16] -> bf16 planes [nsteps][MATH][cp][16] (bmc_split_weight)."""
    nsteps = packed.numel() // (cp * CK)
    out = torch.empty(nsteps * MATH * cp * (CK // 2), device=packed.device, dtype=torch.int32)
    lib.call(lib._split_w, "bmc_split_weight", packed.data_ptr(), out.data_ptr(), nsteps, cp, MATH, _stream())
    return out


def _cache_get(spec: ConvSpec, kind, owner):
    """Packed weights are cached on the ConvSpec, keyed by (kind, id(owner)) and validated by a weak reference to
    the owning parameter plus its version counter (an address or id alone can be recycled by the allocator)."""
    if owner is None:
        return None
    hit = spec._packs.get((kind, id(owner)))
    if hit is not None and hit[0]() is owner and hit[1] == (owner._version, _EPOCH):
        return hit[2]
    return None


def _cache_put(spec: ConvSpec, kind, owner, packed):
    if owner is not None:
        if len(spec._packs) > 64:       # stale entries of dead owners
            for k in [k for k, v in spec._packs.items() if v[0]() is None]:
                del spec._packs[k]
        spec._packs[(kind, id(owner))] = (weakref.ref(owner), (owner._version, _EPOCH), packed)


_STACKS = {}


def invalidate_caches():
    """Drop every cached derived-weight tensor (packed / transposed / plane-split images, stacked group weights, the fused
    chain's weight streams).  The caches are validated by parameter identity + version counter, which every in-place
    torch op and every optimizer step bumps -- but writes through `.data` (p.data.copy_(), p.data.mul_(), EMA / weight-swap
    utilities, the reference's own initialize_weights idiom, models/submodules.py:107-124) do NOT: call this after them.
    load_state_dict() copies with copy_() under no_grad and is tracked."""
    global _EPOCH
    _EPOCH += 1
    _STACKS.clear()
    from . import bie
    bie._CHAIN_CACHE.clear()


_EPOCH = 0      # bumped by invalidate_caches(): part of every pack-cache validation (the packs hang off ConvSpec objects)


def stacked(owners, build, tag=""):
    """A tensor derived from several parameters (the stacked per-group weights of a grouped launch), rebuilt only when
    one of them changed: keyed by `tag` (the layout `build` produces) and their ids, validated by weak references and
    version counters like the pack cache -- and, being one persistent object per parameter set, a valid pack-cache owner
    itself.  (Per step and BIE call this saves two torch.stack launches, a pack and, in the bf16-plane modes, a plane
    split: ~600 tiny launches per step.)"""
    key = (tag,) + tuple(id(o) for o in owners)
    vers = tuple(o._version for o in owners)
    hit = _STACKS.get(key)
    if hit is not None and hit[1] == vers and all(r() is o for r, o in zip(hit[0], owners)):
        return hit[2]
    t = build()
    if len(_STACKS) > 256:
        for k in [k for k, v in _STACKS.items() if any(r() is None for r in v[0])]:
            del _STACKS[k]
    _STACKS[key] = (tuple(weakref.ref(o) for o in owners), vers, t)
    return t


# 3x3 convolutions with 128-granular output channels and enough tiles to fill the chip run through the Winograd transform
# F(2x2, 3x3) (csrc/wino.hip: 16 instead of 36 multiplies per 2x2 output tile and channel pair, fp32 throughout; one
# 256-accumulator workgroup per CU, so small problems stay on the direct kernel).  BMC_WINO=0 switches it off.
WINO = os.environ.get("BMC_WINO", "1") != "0"
WINO_MIN_TILES = int(os.environ.get("BMC_WINO_MIN_TILES", 128))   # tiles (workgroups) from which the F(2x2) kernel beats the direct one: half the CUs (72x80 bs 2: 98.2 -> 94.2 ms against 200; a launch under one round costs one workgroup's serial time whatever its size)


# Round 4: F(4x4, 3x3) (csrc/wino4.hip: 36 instead of 144 multiplies per 4x4 output tile and channel pair -- 1.78x fewer than
# F(2x2) -- fp32 throughout) for every launch that fills the chip with its 16-tile workgroups.  BMC_WINO4=0 keeps F(2x2).
WINO4 = os.environ.get("BMC_WINO4", "1") != "0"
WINO4_MIN_TILES = int(os.environ.get("BMC_WINO4_MIN_TILES", 300))    # workgroup tiles (16 tiles of 4x4 pixels x 128 channels)

# Exact zeros.  Where a pixel's 3x3 receptive field holds nothing, the reference's direct convolution gives EXACTLY its bias;
# with a zero bias that is exactly 0, and relu'(0) = 0 gates the gradient there.  F(2x2) preserves that (every output of its
# minimal algorithm is a combination of products of ITS OWN 3x3 field only), the direct kernel trivially; F(4x4) computes such a
# pixel from a 6x6 patch that also holds its neighbours' data, through rounded transformed weights: a residue of +-1e-8 ... 1e-6
# (it scales with the neighbours' magnitude) instead of 0 -- a coin flip of the ReLU mask.  On dense inputs only the first layers
# of the first window see such fields (round 4's rule, keyed on the caller's `init` flag); on a SPARSE recording with zero biases
# every layer of every window does -- nothing densifies a zero pixel of zero-bias convolutions, LayerNorm2d and per-pixel
# attention -- and bias gradients came out up to 58 % wrong (tests/test_gpu_r5.py::test_sparse_recording_bias_gradients_vs_
# oracle, 0.026 events per pixel at 180x240).  Round 5's rule is keyed on DATA, in two parts:
#   (1) own bias.  A FORWARD 3x3 launch may take F(4x4) if every element of the bias that is added to its result (its own, or for
#       a bias-free launch that adds a residual the bias inside that residual: `rule`) is at least DENSE_FLOOR away from zero: no
#       output of the launch is decided by the residue (bias_dense).
#   (2) dense inputs.  A launch whose input has NO empty receptive field cannot meet the hazard whatever its bias.  The models
#       ARGUE that where they can and say so with the dense_inputs() context: once the input-fusion convolutions -- ReLU layers,
#       evaluated under (1) -- have a bias element >= DENSE_FLOOR each (bias_positive), every pixel of their outputs whose
#       receptive field is EMPTY carries a positive channel, and residual blocks (y = x + ...), LayerNorm2d, 1x1 convolutions
#       and per-pixel attention generically keep a non-zero pixel non-zero: the whole block loop and the tail of that window run
#       inside the context.  This is a HEURISTIC, not a proof: at an OCCUPIED pixel the ReLU can still zero every
#       channel, and LayerNorm2d with a zero beta or a cancellation in x + conv(x) can produce an all-zero pixel, so F(4x4) can
#       still meet an empty 3x3 field with a zero-bias layer inside the block loop.  What it costs then is the coin flip of a few
#       ReLU gates (measured bound: test_sparse_recording_bias_gradients_vs_oracle[trained] 1.3e-4 against the oracle's own
#       float32 floor of 1.2e-4); exact_zero_inputs() / BMC_WINO4=0 are the exact alternatives.
# As `initialize_weights` leaves the biases (zero) every forward launch keeps F(2x2); one optimizer step moves every bias by
# about the learning rate, (2) holds from then on, and only the 4 input-fusion launches of a window depend on (1) -- biases
# cross zero now and then in early training (measured: 8 of 27 vectors had an element within 1e-6 of zero after step 3).  Data
# gradients are never affected (nothing gates on them).  The flags cost one stack of device reductions and ONE host read per
# optimizer step for all 3x3 biases of a model (prime_bias_dense, cached per parameter version).  exact_zero_inputs() forces the
# exact kernels regardless (tests).
_EXACT_ZERO = threading.local()          # .n: nesting depth of exact_zero_inputs on this thread, .d: of dense_inputs
DENSE_FLOOR = 1e-5                       # ~10x the largest residue measured on event-count inputs (1.1e-6)
_DENSE = {}                              # id(bias) -> (weakref, version, min |b|, max b)


class exact_zero_inputs:
    def __enter__(self):
        _EXACT_ZERO.n = getattr(_EXACT_ZERO, "n", 0) + 1

    def __exit__(self, *exc):
        _EXACT_ZERO.n -= 1
        return False


class dense_inputs:
    """The caller vouches that no input of the launches inside has an all-zero pixel (rule (2) above)."""
    def __enter__(self):
        _EXACT_ZERO.d = getattr(_EXACT_ZERO, "d", 0) + 1

    def __exit__(self, *exc):
        _EXACT_ZERO.d -= 1
        return False


def _dense_cached(b):
    hit = _DENSE.get(id(b))
    if hit is not None and hit[0]() is b and hit[1] == b._version:
        return hit
    return None


def prime_bias_dense(biases):
    """min |b| and max b of every bias whose entry is missing or stale: one stack of device reductions, ONE host read.
    (Not while a HIP graph is being captured: entries missing then read as "not dense" -- the exact kernels.)"""
    need = [b for b in biases if b is not None and _dense_cached(b) is None]
    if not need or not need[0].is_cuda or torch.cuda.is_current_stream_capturing():
        return
    vals = torch.stack([torch.stack((b.detach().abs().min(), b.detach().max())) for b in need]).tolist()
    if len(_DENSE) > 512:
        for k in [k for k, v in _DENSE.items() if v[0]() is None]:
            del _DENSE[k]
    for b, (mn, mx) in zip(need, vals):
        _DENSE[id(b)] = (weakref.ref(b), b._version, mn, mx)


def _bias_entry(b):
    hit = _dense_cached(b)
    if hit is None:
        prime_bias_dense([b])
        hit = _dense_cached(b)
    return hit


def bias_dense(b):
    """Is every element of this bias (or of every bias of a tuple) at least DENSE_FLOOR away from zero?  None: no."""
    if b is None:
        return False
    if isinstance(b, (tuple, list)):
        prime_bias_dense(b)
        return all(bias_dense(x) for x in b)
    hit = _bias_entry(b)
    return hit is not None and hit[2] >= DENSE_FLOOR


def bias_positive(b):
    """Does this bias (every bias of a tuple) have an element >= DENSE_FLOOR -- does the ReLU layer it belongs to give every
    pixel a positive channel?"""
    if b is None:
        return False
    if isinstance(b, (tuple, list)):
        prime_bias_dense(b)
        return all(bias_positive(x) for x in b)
    hit = _bias_entry(b)
    return hit is not None and hit[3] >= DENSE_FLOOR


_WINO_ROWS = {}
WINO_MIN_TILES4 = int(os.environ.get("BMC_WINO_MIN_TILES4", 128))   # the same for launches the 4-row tiling takes (a 4-row workgroup: 30 us at 128 channels; the direct kernel's 64-channel tiling of such a launch: 44)


def wino_tiles(B, H, W, cp):
    """-> (workgroup tiles of an F(2x2) launch, rows per tile): 8 x 16 pixels x 128 channels, or 4 x 16 where the launcher picks its
    4-row kernel (csrc/wino.hip::wino_rows: launches that would leave CUs without a tile -- small frames)."""
    key = (B, H, W, cp)
    th = _WINO_ROWS.get(key)
    if th is None:
        th = _WINO_ROWS[key] = lib._wino_rows(B, H, W, cp, 0)
    return B * ((H + th - 1) // th) * ((W + 15) // 16) * (cp // 128), th


def wino_ok(B, H, W, Cout, taps, fwd=False, stride=0, rule=None):
    """Which kernel takes a launch of this geometry?  0: the direct kernel, 2: Winograd F(2x2, 3x3), 4: F(4x4, 3x3).
    (Decided ONCE per launch by the caller and handed to the weight pack and to conv_raw alike: the packed layouts are not
    interchangeable.)  fwd: a forward launch: F(4x4) only inside dense_inputs() or if `rule` -- the bias (tensor, or tuple of
    tensors) that is added to its result -- is dense (the exact-zero rule above).  stride: the widest pixel stride (floats) among the
    launch's sources, residual, mask and output, where it can exceed 512 (channel windows of a wider buffer): the Winograd
    launchers refuse what their 32-bit offsets cannot address, so such a launch must be routed to the direct kernel here."""
    if not WINO or MATH != 0 or taps != 9:
        return 0
    cp = coutpad(Cout)
    if cp % 128:
        return 0
    nt, th = wino_tiles(B, H, W, cp)
    if nt < (WINO_MIN_TILES if th == 8 else min(WINO_MIN_TILES, WINO_MIN_TILES4)):
        return 0
    if H * W * 4 * max(stride, 512) >= 2 ** 31:      # 32-bit per-lane DMA offsets in both Winograd kernels: direct kernel
        return 0
    if WINO4 and W >= 17 and H * W < 2 ** 24:
        n4 = B * ((((H + 3) // 4) * ((W + 3) // 4) + 15) // 16) * (cp // 128)
        if n4 >= WINO4_MIN_TILES and not (fwd and (getattr(_EXACT_ZERO, "n", 0) or
                                                   not (getattr(_EXACT_ZERO, "d", 0) or bias_dense(rule)))):
            return 4
    return 2


PAIR_SMALL = os.environ.get("BMC_PAIR_SMALL", "1") != "0"
PAIR_ALWAYS = os.environ.get("BMC_PAIR_SMALL") == "2"      # experiments: pair at every size
PAIR_BELOW_TILES = 200        # 8 x 16-pixel tiles of one block's launch under which the pair shares a launch


def pair_small(B2, H, W):
    """Small frames: should two weight-distinct, equally shaped, independent residual blocks (ParallelBlk.conv1 / conv1_st,
    models/BMCNet.py:19-22) run as ONE two-group launch per convolution over the stacked inputs?  Yes when one block's launch
    does not reach a workgroup per CU but the pair does (the pair's 3x3 launches then take the Winograd kernel in fp32): at
    31x56, bs 4 the step's 3x3 convolutions were 128-tile launches at 86 TFLOP/s.  The price is one concatenation of the two
    inputs -- negligible at these sizes, which is why large frames keep the separate launches."""
    t = B2 * ((H + 7) // 8) * ((W + 15) // 16)
    return PAIR_SMALL and WINO and MATH == 0 and (PAIR_ALWAYS or t < PAIR_BELOW_TILES <= 2 * t)      # (bf16 arithmetic, no Winograd: 72 -> 76 ms, not used)


def _packed_weight(w4: torch.Tensor, spec: ConvSpec, owner, wino=False):
    """[G,Cout,Cin,taps] -> MFMA staging layout (bmc_pack_weight; wino: the transformed weights of bmc_pack_weight_wino).
    owner: the parameter tensor w4 was derived from (cache key), or None for no caching."""
    G, Cout, Cin, taps = w4.shape
    if wino:
        npos = 36 if wino == 4 else 16
        hit = _cache_get(spec, ("fw", npos), owner)
        if hit is not None:
            return hit
        cp = coutpad(Cout)
        out = torch.empty(G * spec.kpad * npos * cp, device=w4.device, dtype=torch.float32)
        fn, nm = (lib._pack_wino4, "bmc_pack_weight_wino4") if wino == 4 else (lib._pack_wino, "bmc_pack_weight_wino")
        lib.call(fn, nm, w4.data_ptr(), spec.kmap(w4.device).data_ptr(), G, Cout, Cin,
                 spec.kpad, cp, 0, 0, 0, out.data_ptr(), _stream())
        _cache_put(spec, ("fw", npos), owner, out)
        return out
    hit = _cache_get(spec, ("f", MATH), owner)
    if hit is not None:
        return hit
    cp = coutpad(Cout)
    out = torch.empty(G * spec.kpad * taps * cp, device=w4.device, dtype=torch.float32)
    lib.call(lib._pack_w, "bmc_pack_weight", w4.data_ptr(), spec.kmap(w4.device).data_ptr(), G, Cout, Cin, taps,
             spec.kpad, cp, out.data_ptr(), _stream())
    if MATH:
        out = _split_planes(out, cp)
    _cache_put(spec, ("f", MATH), owner, out)
    return out


def _packed_weight_t(w4: torch.Tensor, spec: ConvSpec, src_index: int, owner, wino=False):
    G, Cout, Cin, taps = w4.shape
    k0 = sum(spec.nch[:src_index])
    nk = spec.nch[src_index]
    nkpad = coutpad(nk)
    c16 = round_up(Cout, CK)
    if wino:
        npos = 36 if wino == 4 else 16
        hit = _cache_get(spec, ("tw", src_index, npos), owner)
        if hit is not None:
            return hit
        out = torch.empty(G * c16 * npos * nkpad, device=w4.device, dtype=torch.float32)
        fn, nm = (lib._pack_wino4, "bmc_pack_weight_wino4") if wino == 4 else (lib._pack_wino, "bmc_pack_weight_wino")
        lib.call(fn, nm, w4.data_ptr(), spec.kmap(w4.device).data_ptr(), G, Cout, Cin,
                 c16, nkpad, 1, k0, nk, out.data_ptr(), _stream())
        _cache_put(spec, ("tw", src_index, npos), owner, out)
        return out
    hit = _cache_get(spec, ("t", src_index, MATH), owner)
    if hit is not None:
        return hit
    out = torch.empty(G * c16 * taps * nkpad, device=w4.device, dtype=torch.float32)
    lib.call(lib._pack_wt, "bmc_pack_weight_t", w4.data_ptr(), spec.kmap(w4.device).data_ptr(), G, Cout, Cin, taps,
             k0, nk, nkpad, c16, out.data_ptr(), _stream())
    if MATH:
        out = _split_planes(out, nkpad)
    _cache_put(spec, ("t", src_index, MATH), owner, out)
    return out



# --------------------------------------------------------------------------
# raw launches
# --------------------------------------------------------------------------
def conv_raw(srcs: List[lib.Src], wpacked, w_group_stride, bias, bias_group_stride, out_ptr, out_batch_stride,
             out_pix_stride, B, H, W, Cout, taps, relu=False, residual: Optional[lib.Src] = None, bpg=None,
             accumulate=False, flops=0.0, mask: Optional[lib.Src] = None, wino=False):
    """wino (0 / 2 / 4): wpacked is the Winograd pack of F(2x2) / F(4x4) (the caller decided with wino_ok() and packed
    accordingly); w_group_stride is still given for the direct layout ([K/16][9 taps][Coutpad][16]) and converted here (16 /
    36 positions instead of 9 taps)."""
    a = lib.ConvArgs()
    a.nsrc = len(srcs)
    for i, s in enumerate(srcs):
        a.src[i] = s
    a.wpacked = wpacked.data_ptr()
    a.bias = bias.data_ptr() if bias is not None else None
    a.w_group_stride = w_group_stride * MATH // 2 if MATH else w_group_stride   # floats, or dwords of bf16 planes
    a.math = MATH
    if wino:
        assert MATH == 0 and taps == 9
        a.w_group_stride = w_group_stride // 9 * (36 if wino == 4 else 16)
        a.math = 5 if wino == 4 else 4          # BMC_MATH_FP32_WINO4 / BMC_MATH_FP32_WINO
    a.bias_group_stride = bias_group_stride
    a.batch_per_group = bpg if bpg else B
    a.out = out_ptr
    a.out_batch_stride = out_batch_stride
    a.out_pix_stride = out_pix_stride
    a.B, a.H, a.W = B, H, W
    a.Cout, a.Coutpad = Cout, coutpad(Cout)
    a.taps = taps
    a.relu = int(relu)
    a.residual = residual if residual is not None else _null_src()
    a.mask = mask if mask is not None else _null_src()
    a.accumulate = int(accumulate)
    e0 = _prof_begin()
    lib.call(lib._conv, "bmc_conv", C.byref(a), _stream())
    _prof_end(e0, ("wino4_conv<9,128>" if wino == 4 else "wino_conv<9,128>") if wino else
              "conv_kernel<%d,%d>" % (taps, 32 if a.Coutpad == 32 else 128), flops,
              4.0 * B * H * W * (sum(x.nch for x in srcs) + Cout * (1 + (residual is not None) + (mask is not None) + bool(accumulate))))
    if PROFILE is not None and e0 is not None:
        PROFILE_WINO[0 if wino else 1] += 1


_ZEROS = {}


def _zeros(device):
    z = _ZEROS.get(device)
    if z is None:
        z = torch.zeros(64, device=device, dtype=torch.float32)
        _ZEROS[device] = z
    return z


TAP_SPLIT_TILES = 8


def pgemm_raw(a_src: lib.Src, srcs: List[lib.Src], B, H, W, taps, bpg, M, N, device, flops=0.0, want_bias=False):
    """Returns (slabs, nsplit, G), or with want_bias (slabs, nsplit, G, bias_slabs): per-workgroup column sums of the
    A operand (bias-gradient partials) for bmc_pgemm_reduce_weight."""
    G = B // bpg
    mpad, npad = round_up(M, 32), round_up(N, 32)
    if taps == 9:       # one 8-wave workgroup per CU, 64 columns per workgroup (32 in the bf16x6 mode: pgemm_bf.hip)
        cols = 32 if MATH == 3 else 64
        n_nblk = (npad + cols - 1) // cols
        tiles = ((H + 3) // 4) * ((W + 15) // 16)
        target = 256
    else:               # same kernel family, 128 columns per workgroup
        n_nblk = (npad + 127) // 128
        tiles = (H * W + 63) // 64
        target = 256
    other = G * ((mpad + 127) // 128) * n_nblk
    nsplit = max(1, min(bpg * tiles, target // max(other, 1)))
    # small images: fewer than TAP_SPLIT_TILES pixel tiles per workgroup -> one tap row per workgroup, a third of the splits
    # (every split writes a whole slab: at 31x56 the slab writes cost as much as the MFMAs; bmc_pgemm_args_t.tap_groups)
    tap_groups = 1
    if taps == 9 and MATH in (0, 1) and bpg * tiles < TAP_SPLIT_TILES * nsplit and nsplit >= 3:
        tap_groups = 3
        nsplit = max(1, min(bpg * tiles, target // max(other * 3, 1)))
    slabs = torch.empty(nsplit * G * taps * mpad * npad, device=device, dtype=torch.float32)
    p = lib.PgemmArgs()
    p.a = a_src
    p.nsrc = len(srcs)
    for i, s in enumerate(srcs):
        p.src[i] = s
    p.B, p.H, p.W, p.taps = B, H, W, taps
    p.batch_per_group = bpg
    p.slabs = slabs.data_ptr()
    p.nsplit = nsplit
    p.zeros = _zeros(device).data_ptr()
    p.math = MATH
    p.tap_groups = tap_groups
    bslabs = None
    if want_bias:
        bslabs = torch.empty(nsplit * G * 4 * mpad, device=device, dtype=torch.float32)
        p.bias_slabs = bslabs.data_ptr()
    else:
        p.bias_slabs = None
    _on_side(slabs, bslabs)
    e0 = _prof_begin()
    lib.call(lib._pgemm, "bmc_pgemm", C.byref(p), _stream())
    _prof_end(e0, "pgemm_kernel<%d>" % taps, flops, 4.0 * B * H * W * (a_src.nch + sum(x.nch for x in srcs)) + 4.0 * slabs.numel())
    if want_bias:
        return slabs, nsplit, G, bslabs
    return slabs, nsplit, G


# Weight gradients of the dense 3x3 128 -> 128 convolutions (the residual blocks: 88 % of the 3x3 weight-gradient work) go
# through the Winograd transform too (csrc/wino_wgrad.hip: 16 instead of 36 multiplies per 2x2 tile and channel pair; fp32,
# deterministic).  BMC_WINO_WGRAD=0 (or BMC_WINO=0) leaves them to the pixel-reduction GEMM.
WINO_WGRAD = os.environ.get("BMC_WINO_WGRAD", "1") != "0"
# Round 5: the same through F(4x4, 3x3) (csrc/wino4_wgrad.hip: 36 multiplies per 4x4 tile and channel pair, 1.78x fewer than
# F(2x2)).  Built, bit-for-bit deterministic, 2.9e-6 against float64 -- and NOT faster on this chip: 0.377 ms against F(2x2)'s
# 0.382 ms for the 8-image launch at 180x240 (0.44 against 0.40 with the reduction).  Both of its operands are transformed per
# tile, so every workgroup moves 39 KB through the CU's vector-memory path per 2 304 matrix-pipe cycles (that path takes 1 KB per
# 40-60 cycles: PMC + ablations) and runs ~100 vector instructions per wave and stage, each of which waits behind the fp32 MFMAs
# the SIMD partner has queued (in-kernel stamps: the transform phase of a wave takes 2 100 cycles for ~60 instructions).
# NOTEBOOK.md, "F(4x4) weight gradient", has the numbers.  Off by default; BMC_WINO4_WGRAD=1 routes launches with at least
# WINO4_WGRAD_MIN_STAGES stages (4 tiles each) per workgroup to it (the tests do).
WINO4_WGRAD = os.environ.get("BMC_WINO4_WGRAD", "0") != "0"
WINO4_WGRAD_MIN_STAGES = int(os.environ.get("BMC_WINO4_WGRAD_MIN_STAGES", 24))


def wgrad_wino4_ok(B, H, W):
    if not WINO4_WGRAD:
        return False
    stages = B * ((H + 3) // 4) * (((W + 3) // 4 + 3) // 4)
    return stages >= 32 * WINO4_WGRAD_MIN_STAGES


def wino_wgrad_ok(a_src, x_srcs, spec, taps, Cout, G):
    """Does this weight gradient take the Winograd kernel?  One dense 128-channel source (pix_stride 128) whose channels are a
    contiguous column range [k0, k0 + 128) of the weight tensor, 128 output channels, one weight group, fp32 arithmetic."""
    if not (WINO and WINO_WGRAD and MATH == 0 and taps == 9 and G == 1 and Cout == 128 and len(x_srcs) == 1 and spec.kpad == 128):
        return False
    xs = x_srcs[0]
    if not (xs.nch == 128 and xs.pix_stride == 128 and a_src.nch == 128 and a_src.pix_stride == 128):
        return False
    k0 = spec.kmap_host[0]
    return k0 >= 0 and spec.kmap_host == list(range(k0, k0 + 128))


# Merged weight gradients (round 5).  `para_reschunk` holds ONE ParallelBlk n_b times (models/BMCNet.py:19-32), so every 3x3 weight of
# the block loop is used n_b times per window, and backward meets those uses one after the other: n_b launches of the Winograd
# weight-gradient kernel, each with its own set of partial sums (64 MB written and read again whatever the image size) and its own
# reduction, all adding into the same .grad.  The sum over uses is a sum over more images: inside a backward pass the uses of a
# sink parameter are QUEUED (operand descriptors + references that keep the operands alive) and every WGRAD_MERGE of them leave as
# ONE launch over several (dY, x) pairs (bmc_wgrad_wino_multi) with one reduction.  Queues that are not full are flushed when the
# backward pass ends (engine callback, queued before the side stream's join) and by wgrad_join() -- i.e. before anything reads a
# .grad.  Uses of different windows never share a launch (next_window(): the groups must not depend on how backward is cut into
# passes).  Costs: at most WGRAD_MERGE - 1 transient gradients per weight stay alive a little longer (31x56 bs 4: peak memory unchanged, 6.4 GiB).
# BMC_WGRAD_MERGE=1 switches it off.  Deterministic (the queue order is backward's order); NOT bit-identical to the unmerged
# order of summation.
WGRAD_MERGE = max(1, min(8, int(os.environ.get("BMC_WGRAD_MERGE", 5))))
# ... for uses of at most this many pixels: what merging saves is per-launch overhead (partial sums, reduction, ramp and drain),
# 31x56 bs 4: 82.8 -> 74.4 ms per step, 45x80 bs 2: 82.6 -> 73.6; at C2 (345 600 pixels per use) the launches are long enough and
# five of them in one cost the overlap with the data-gradient chain more than they save: 722.2 -> 726.1 ms
WGRAD_MERGE_MAX_PIXELS = int(os.environ.get("BMC_WGRAD_MERGE_MAX_PIXELS", 1 << 17))


_WINDOW = [0]                # id of the recurrent window whose forward is running (models: next_window() per Backbone forward)


def next_window():
    """Called by the models at the top of every window's forward: weight-gradient uses are merged within a window only, so the
    launches are the same whether a window's backward runs in the step's one backward pass or (per-window recompute) from a
    checkpoint's recomputed graph -- and the two stay bit-identical."""
    _WINDOW[0] += 1
    return _WINDOW[0]


def current_window():
    return _WINDOW[0]


class _MergeQueue:
    __slots__ = ("items", "spec", "dev", "w_param", "b_param", "want_bias", "k0", "full", "H", "W", "window", "pg")

    def __init__(self, spec, dev, w_param, b_param, want_bias, k0, full, H, W, window, pg=None):
        self.items = []
        self.spec, self.dev, self.w_param, self.b_param, self.want_bias, self.k0, self.full, self.H, self.W, self.window, self.pg = \
            spec, dev, w_param, b_param, want_bias, k0, full, H, W, window, pg      # pg: (taps, Cout, w_shape) of a pixel-reduction queue


# Queues belong to the autograd graph task (backward pass) that filled them: a nested pass -- torch.utils.checkpoint(use_reentrant=True),
# a Function whose backward calls autograd.backward -- has its own task id, its own queues and its own end-of-pass flush, and neither
# launches nor drops what the pass around it has queued.  Queues of a pass that RAISED (it never flushes) are dropped by wgrad_join()
# once no backward pass is running.
_MERGE = {}                  # graph task id -> {(id(w_param), k0, H, W, want_bias) -> _MergeQueue}
_FLUSH_QUEUED = set()        # graph tasks whose end-of-pass flush is queued
PTR_TABLE_MAX = 256          # csrc/stream_ops.hip: bmc_ptr_table takes its pointers by value in the argument block


def flush_wgrads(task=None):
    """Launch every weight gradient queued by the graph task `task` (default: the running one); idempotent."""
    if task is None:
        task = torch._C._current_graph_task_id()
    _FLUSH_QUEUED.discard(task)
    qs = _MERGE.pop(task, None)
    if not qs:
        return
    for q in list(qs.values()):
        if q.items:
            _launch_merged(q)


def _queue_flush(task):
    """The end-of-pass flush of graph task `task`, queued once (the engine runs a task's callbacks in the order they were queued)."""
    if task not in _FLUSH_QUEUED:
        _FLUSH_QUEUED.add(task)
        torch.autograd.Variable._execution_engine.queue_callback(lambda: flush_wgrads(task))


_TABLES = {}      # (stream, image pointers) -> device table holding exactly those pointers


def _table_src(srcs, batches, btot, dev):
    """One operand of a merged pixel-reduction launch: the images of `srcs` (one lib.Src per use, batches[i] images each) behind
    ONE descriptor in BMC_SRC_TABLE mode.  A device table is a pure function of the pointers it holds, and in steady-state training
    the caching allocator hands out the same addresses step after step: tables are memoised by (stream, pointers) -- the stream
    because a table is written by a kernel, and only launches of the same stream are ordered behind that write -- so that from the
    second step on a merged launch costs no `bmc_ptr_table` launch (31x56: 130 launches per step in fp32, 298 in bf16)."""
    ptrs = []
    for s_, nb in zip(srcs, batches):
        mod = s_.batch_mod if s_.batch_mod >= 1 else 1
        ptrs.extend(s_.ptr + 4 * (((i + s_.batch_shift) % mod) * s_.batch_stride) for i in range(nb))
    st = _stream()
    key = (dev.index, st.value if hasattr(st, "value") else st, tuple(ptrs))
    table = _TABLES.get(key)
    if table is None:
        if len(_TABLES) >= 8192:
            _TABLES.clear()
        table = torch.empty(btot, device=dev, dtype=torch.int64)
        lib.call(lib._ptr_table, "bmc_ptr_table", (C.c_ulonglong * btot)(*ptrs), btot, table.data_ptr(), st)
        if not torch.cuda.is_current_stream_capturing():      # (a table made during a graph capture belongs to that graph's pool)
            _TABLES[key] = table
    t = lib.Src()
    t.ptr = table.data_ptr()
    t.batch_stride, t.pix_stride, t.nch, t.batch_shift, t.batch_mod = 0, srcs[0].pix_stride, srcs[0].nch, 0, -1      # BMC_SRC_TABLE
    return t, table


def _launch_merged(q):
    items, q.items = q.items, []
    keep = tuple(t for it in items for t in it[3])
    npx = items[0][2] * q.H * q.W          # (the side-stream decision of a pass is made per launch size as before: first segment)
    if q.pg is not None:
        taps, Cout, w_shape = q.pg
        batches = [it[2] for it in items]
        btot = sum(batches)
        nx = len(items[0][1])
        with wgrad_side(npx, [q.w_param, q.b_param] if q.want_bias else [q.w_param], keep):
            a_t, tab_a = _table_src([it[0] for it in items], batches, btot, q.dev)
            x_tt = [_table_src([it[1][k] for it in items], batches, btot, q.dev) for k in range(nx)]
            x_t = [t for t, _ in x_tt]
            _on_side(tab_a, *[tb for _, tb in x_tt])
            r = pgemm_raw(a_t, x_t, btot, q.H, q.W, taps, btot, Cout, q.spec.kpad, q.dev,
                          flops=2.0 * btot * q.H * q.W * Cout * taps * q.spec.kreal, want_bias=q.want_bias)
            reduce_wgrad(r[0], r[1], 1, taps, Cout, q.spec, q.dev, r[3] if q.want_bias else None, q.w_param,
                         q.b_param if q.want_bias else None, w_shape)
        return
    with wgrad_side(npx, [q.w_param, q.b_param] if q.want_bias else [q.w_param], keep):
        _wgrad_wino([it[0] for it in items], [it[1] for it in items], [it[2] for it in items], q.H, q.W, q.spec, q.dev, q.w_param,
                    q.b_param, None, q.want_bias, q.k0, q.full)


def _queue_use(key, make_queue, item, task, npx):
    """Append one use to its merge queue (created by make_queue() if new / stale) and launch the queue when it is full."""
    qs = _MERGE.setdefault(task, {})
    q = qs.get(key)
    if q is None or q.w_param is not item[4] or q.window != item[5]:       # (another window's uses: the old queue leaves first)
        if q is not None and q.items:
            _launch_merged(q)
        q = qs[key] = make_queue()
    if q.pg is not None and q.items and sum(it[2] for it in q.items) + item[2] > PTR_TABLE_MAX:
        _launch_merged(q)            # (a pointer table holds PTR_TABLE_MAX images: what is queued leaves before this use joins)
    _queue_flush(task)               # the first queued use of this backward pass: flush what is left when the pass ends
    _side_arm(npx)                   # (the pass's side-stream decision is made by its first USE, as without the queue)
    q.items.append(item[:4])
    if len(q.items) >= WGRAD_MERGE:
        _launch_merged(q)


def _mergeable(B, H, W, w_param, b_param, want_bias, keep, window):
    task = torch._C._current_graph_task_id()
    ok = (WGRAD_MERGE > 1 and window is not None and B * H * W <= WGRAD_MERGE_MAX_PIXELS and task >= 0 and w_param is not None
          and not isinstance(w_param, (tuple, list)) and is_sink(w_param) and (not want_bias or is_sink(b_param)) and keep
          and not torch.cuda.is_current_stream_capturing())
    return ok, task


def wgrad_pgemm(a_src, x_srcs, B, H, W, taps, Cout, spec, dev, w_param, b_param, w_shape, G=1, want_bias=True, keep=(), window=None):
    """Weight (+ bias) gradient through the pixel-reduction GEMM + its slab reduction, with reduce_wgrad's conventions.  Uses of one
    sink parameter inside a window are queued and merged like wgrad_wino's: the merged launch reads its operands through
    per-image pointer tables (bmc_src_t BMC_SRC_TABLE; bmc_ptr_table)."""
    if G == 1:
        ok, task = _mergeable(B, H, W, w_param, b_param, want_bias, keep, window)
        if ok and B <= PTR_TABLE_MAX and all(x.batch_mod != -1 for x in x_srcs) and a_src.batch_mod != -1:
            key = (id(w_param), "pg", taps, len(x_srcs), H, W, bool(want_bias))
            _queue_use(key, lambda: _MergeQueue(spec, dev, w_param, b_param, want_bias, None, spec.covers_all, H, W, window,
                                                pg=(taps, Cout, w_shape)),
                       (a_src, list(x_srcs), B, tuple(keep), w_param, window), task, B * H * W)
            return None, None
    with wgrad_side(B * H * W, _flat_params(w_param, b_param if want_bias else None), keep):
        r = pgemm_raw(a_src, x_srcs, B, H, W, taps, B // G, Cout, spec.kpad, dev, flops=2.0 * B * H * W * Cout * taps * spec.kreal,
                      want_bias=want_bias)
        return reduce_wgrad(r[0], r[1], G, taps, Cout, spec, dev, r[3] if want_bias else None, w_param,
                            b_param if want_bias else None, w_shape)


def wgrad_wino(a_src, x_src, B, H, W, spec, dev, w_param, b_param, w_shape, want_bias=True, k0=None, full=None, keep=(), window=None):
    """(dW, db) of a convolution wino_wgrad_ok() accepted, with reduce_wgrad's conventions: None where the sums went straight
    into a leaf parameter's .grad.  k0 / full: the launch covers only the weight columns [k0, k0 + 128) of a wider convolution
    (one 128-channel source of a multi-source launch: split_wgrad_ok).  window: the recurrent window the use belongs to
    (current_window() at its forward) -- uses are merged within one window; None: not merged."""
    ok, task = _mergeable(B, H, W, w_param, b_param, want_bias, keep, window)
    if ok and not wgrad_wino4_ok(B, H, W):
        k0_ = spec.kmap_host[0] if k0 is None else k0
        key = (id(w_param), k0_, H, W, bool(want_bias))
        _queue_use(key, lambda: _MergeQueue(spec, dev, w_param, b_param, want_bias, k0_, spec.covers_all if full is None else full, H, W,
                                            window),
                   (a_src, x_src, B, tuple(keep), w_param, window), task, B * H * W)
        return None, None
    with (wgrad_side(B * H * W, [w_param, b_param] if want_bias else [w_param], keep) if w_param is not None else _NOCTX):
        return _wgrad_wino(a_src, x_src, B, H, W, spec, dev, w_param, b_param, w_shape, want_bias, k0, full)


def _wgrad_wino(a_src, x_src, B, H, W, spec, dev, w_param, b_param, w_shape, want_bias, k0, full):
    # a_src / x_src / B: one operand pair, or lists of them (a merged launch: several uses of one weight)
    multi = isinstance(a_src, (list, tuple))
    Bt = sum(B) if multi else B
    f4 = not multi and wgrad_wino4_ok(B, H, W)
    if f4:
        f_ns, f_main, f_red, nm, npos, kind = lib._ww4_nsplit, lib._ww4, lib._ww4_red, "bmc_wgrad_wino4", 36, "wgrad_wino4<9>"
    else:
        f_ns, f_main, f_red, nm, npos, kind = lib._ww_nsplit, lib._ww, lib._ww_red, "bmc_wgrad_wino", 16, "wgrad_wino<9>"
    nsplit = f_ns(Bt, H, W)
    part = torch.empty(nsplit * npos * 128 * 128, device=dev, dtype=torch.float32)
    bpart = torch.empty(nsplit * 128, device=dev, dtype=torch.float32) if want_bias else None
    _on_side(part, bpart)
    e0 = _prof_begin()
    if multi:
        n = len(a_src)
        lib.call(lib._ww_multi, "bmc_wgrad_wino_multi", (lib.Src * n)(*a_src), (lib.Src * n)(*x_src), (C.c_int * n)(*B), n, H, W, nsplit,
                 part.data_ptr(), bpart.data_ptr() if want_bias else None, _stream())
    else:
        lib.call(f_main, nm, C.byref(a_src), C.byref(x_src), B, H, W, nsplit, part.data_ptr(),
                 bpart.data_ptr() if want_bias else None, _stream())
    _prof_end(e0, kind, 2.0 * Bt * H * W * 128 * 9 * 128, 4.0 * Bt * H * W * 256 + 4.0 * part.numel())
    k0 = spec.kmap_host[0] if k0 is None else k0
    full = spec.covers_all if full is None else full
    sg = sink_group([w_param, b_param] if want_bias else [w_param], full) if w_param is not None else None
    if sg is not None:
        (gw, *rest), acc = sg
        lib.call(f_red, nm + "_reduce", part.data_ptr(), nsplit, gw.data_ptr(), spec.cin, k0, acc,
                 bpart.data_ptr() if want_bias else None, rest[0].data_ptr() if want_bias else None, _stream())
        return None, None
    assert not multi
    dw = (torch.empty if full else torch.zeros)(128 * spec.cin * 9, device=dev, dtype=torch.float32)
    db = torch.empty(128, device=dev, dtype=torch.float32) if want_bias else None
    lib.call(f_red, nm + "_reduce", part.data_ptr(), nsplit, dw.data_ptr(), spec.cin, k0, 0,
             bpart.data_ptr() if want_bias else None, db.data_ptr() if want_bias else None, _stream())
    return dw.view(w_shape), db


def split_wgrad(spec, metas):
    """A multi-source 3x3 convolution with 128 outputs whose weight gradient is worth splitting: every dense 128-channel
    source (contiguous weight columns, no batch map) takes the Winograd kernel, the narrow rest ONE pixel-reduction launch
    over a sub-spec of the same weight tensor (the shared part of conv_fs: two 128-channel sources + 2 x 16 channels;
    conv_fpst, conv_fps: one + 16 / 32).  -> ([(source index, k0)], [narrow source indices], sub-spec or None) or None.
    Cached on the spec."""
    cache = spec.__dict__.setdefault("_split", {})
    key = tuple(tuple(m) for m in metas)
    if key in cache:
        return cache[key] or None
    big, rest, off = [], [], 0
    for i, n in enumerate(spec.nch):
        seg = spec.kmap_host[off:off + n]
        c0, nch, shift, mod, b0 = metas[i]
        if n == 128 and nch == 128 and shift == 0 and mod is None and seg[0] >= 0 and seg == list(range(seg[0], seg[0] + 128)):
            big.append((i, seg[0]))
        else:
            rest.append(i)
        off += n
    res = False
    if big and len(spec.nch) > 1:
        sub = None
        if rest:
            offs = [sum(spec.nch[:i]) for i in range(len(spec.nch))]
            sub = ConvSpec([spec.kmap_host[offs[i]:offs[i] + spec.nch[i]] for i in rest], cin=spec.cin)
        res = (big, rest, sub)
    cache[key] = res
    return res or None


def relu_bwd(dy, y):
    g = torch.empty_like(dy)
    lib.call(lib._relu_bwd, "bmc_relu_bwd", dy.data_ptr(), y.data_ptr(), g.data_ptr(), dy.numel(), _stream())
    return g


def group_sum(g, groups):
    """[groups*n, ...] -> [n, ...]: sum over the batch groups, fixed order (bmc_group_sum)."""
    n = g.shape[0] // groups
    out = torch.empty((n,) + tuple(g.shape[1:]), device=g.device, dtype=torch.float32)
    lib.call(lib._group_sum, "bmc_group_sum", g.data_ptr(), groups, out.numel(), out.data_ptr(), _stream())
    return out


def colsum(x2d_ptr, npix, pix_stride, Cn, device):
    ws = torch.empty(2048 * Cn, device=device, dtype=torch.float32)
    out = torch.empty(Cn, device=device, dtype=torch.float32)
    lib.call(lib._colsum, "bmc_colsum", x2d_ptr, npix, pix_stride, Cn, ws.data_ptr(), out.data_ptr(), 0, _stream())
    return out


# --------------------------------------------------------------------------
# parameter gradients: accumulate in the slab reduction, not in autograd
# --------------------------------------------------------------------------
# Every parameter of the network is used 10-40 times per recurrent window (5 weight-shared blocks, twin branches) and in
# 8 windows per step: routed through autograd, each use costs one slab reduction into a fresh tensor plus one ATen add
# into the running sum (~1 800 add launches per C2 step).  When the weight of a launch IS a leaf parameter, its reduction
# adds straight into param.grad instead (bmc_pgemm_reduce_weight(accumulate=1), fixed order = backward's execution order,
# deterministic) and autograd gets None for it.  Derived weights (slices / stacks of parameters) keep the autograd route.
# Side effect to know about: post-accumulate-grad hooks do not fire for such parameters (parallel.GradAllReducer stages
# the finished gradients in its optimizer pre-step hook instead).  BMC_ACCUM_GRADS=0 switches this off.
ACCUM_PARAM_GRADS = os.environ.get("BMC_ACCUM_GRADS", "1") != "0"


def is_sink(p):
    """May a launch add its weight gradient straight into p.grad?  Only for contiguous leaf parameters, and only when nobody
    is listening on autograd for them: a parameter with tensor hooks or post-accumulate-grad hooks keeps the autograd
    route (the hooks fire only when autograd accumulates), unless the hook's owner declared that it stages sink gradients
    itself (parallel.GradAllReducer records the ids of its own hooks in p._bmc_sink_hooks: only those may be present -- a
    clipping or logging hook registered beside them, before or after, or the hook of a reducer that was dropped without
    detach(), sends the parameter back to the autograd route).  Not detectable from here, hence documented in INTEGRATION.md:
    torch DDP / FSDP (hooks on the AccumulateGrad nodes) and torch.autograd.grad(loss, params) need BMC_ACCUM_GRADS=0 /
    set_accumulate_param_grads(False)."""
    if not (ACCUM_PARAM_GRADS and p is not None and p.is_leaf and p.requires_grad and p.is_contiguous()):
        return False
    if p._backward_hooks:
        return False
    post = p._post_accumulate_grad_hooks
    if post and not set(post.keys()) <= getattr(p, "_bmc_sink_hooks", _NO_HOOKS):      # a hook that is not a sink-aware owner's
        return False
    return True


_NO_HOOKS = frozenset()


def set_accumulate_param_grads(on: bool):
    """Switch the direct-to-.grad route for leaf parameters (default on; BMC_ACCUM_GRADS=0 at import time switches it off)."""
    global ACCUM_PARAM_GRADS
    ACCUM_PARAM_GRADS = bool(on)


def sink_group(params, full=True):
    """params: the parameters ONE launch writes gradients for (it has one accumulate flag for all of them).
    -> ([their .grad tensors], accumulate) if every one is a leaf parameter, else None.  First use in a step allocates
    the accumulators (accumulate = 0: the launch overwrites them); mixed states are aligned by zero-filling.
    full=False: the launch defines only part of the gradient (ConvSpec.covers_all is false): new accumulators start as zeros."""
    if not all(is_sink(p) for p in params):
        return None
    if _SIDE and torch._C._current_graph_task_id() < 0:
        wgrad_join()        # outside a backward pass: nothing will run the join a raised backward left behind
    for p in params:        # seen by parallel.GradAllReducer.finish(): this gradient did not (only) come through autograd
        p._bmc_sink_touched = True
    missing = [p.grad is None for p in params]
    if all(missing) and full:
        for p in params:
            p.grad = torch.empty_like(p, memory_format=torch.contiguous_format)
        acc = 0
    else:
        for p, m in zip(params, missing):
            if m:
                p.grad = torch.zeros_like(p, memory_format=torch.contiguous_format)
            elif not p.grad.is_contiguous():
                p.grad = p.grad.contiguous()
        acc = 1
        if any(missing) and _STREAM_OVERRIDE is not None:
            _SIDE[_cur_dev()].follow_main()         # the zero fill ran on the launch stream: the side stream's adds come after it
    return [p.grad for p in params], acc


# --------------------------------------------------------------------------
# batched C x C products (csrc/smallmm.hip): the BIE's attention without the value tensor (bie.py)
# --------------------------------------------------------------------------
def small_mm(terms, nbatch, bpg, M, N, K, c=None, alpha=1.0, uv=None, vec_out=None, accumulate=False):
    """C[b][i][j] (=|+=) alpha (sum_t A_t[b] B_t[b] + u[b] v[b]^T), vec[b][i] (=|+=) alpha sum_t A_t[b] w_t[b]  (include/bmc_hip.h:
    bmc_small_mm).  Operands are (tensor, batch stride, group stride, stride, stride) tuples in floats -- tensor may be a view, its
    storage offset is honoured:
        terms:   [(A, B, w or None), ...]   A = (t, sb, sg, s_i, s_k), B = (t, sb, sg, s_k, s_j), w = (t, sb, sg, s_k)
        c:       (t, sb, sg, s_i, s_j) or None;   uv: ((u, sb, sg), (v, sb, sg)) or None;   vec_out: (t, sb, sg) or None"""
    a = lib.SmallMmArgs()
    a.nterms = len(terms)
    for i, (A, B, w) in enumerate(terms):
        t = a.t[i]
        t.a, t.a_sb, t.a_sg, t.a_si, t.a_sk = A[0].data_ptr(), A[1], A[2], A[3], A[4]
        t.b, t.b_sb, t.b_sg, t.b_sk, t.b_sj = B[0].data_ptr(), B[1], B[2], B[3], B[4]
        if w is not None:
            t.w, t.w_sb, t.w_sg, t.w_sk = w[0].data_ptr(), w[1], w[2], w[3]
    a.nbatch, a.batch_per_group, a.M, a.N, a.K, a.alpha = nbatch, bpg, M, N, K, alpha
    if uv is not None:
        (u, usb, usg), (v, vsb, vsg) = uv
        a.u, a.u_sb, a.u_sg, a.v, a.v_sb, a.v_sg = u.data_ptr(), usb, usg, v.data_ptr(), vsb, vsg
    if c is not None:
        a.c, a.c_sb, a.c_sg, a.c_si, a.c_sj = c[0].data_ptr(), c[1], c[2], c[3], c[4]
    if vec_out is not None:
        a.vec_out, a.vo_sb, a.vo_sg = vec_out[0].data_ptr(), vec_out[1], vec_out[2]
    a.accumulate = int(accumulate)
    lib.call(lib._small_mm, "bmc_small_mm", C.byref(a), _stream())


# --------------------------------------------------------------------------
# weight gradients beside the data-gradient chain
# --------------------------------------------------------------------------
# Nothing in backward waits for a weight gradient: it only has to be in .grad when the optimizer steps.  So the weight-gradient
# kernels (pixel-reduction GEMMs, the Winograd weight gradient, their reductions: a quarter to a third of the step) run on a
# SIDE stream beside the convolutions of the data-gradient chain instead of between them.  Every kernel fills the 256 CUs,
# but in ROUNDS of one tile per CU -- the F(4x4) convolution's 1 352 workgroup tiles of a 2B launch at 180x240 are 5.28 rounds,
# so 186 CUs idle through the sixth, and every dependent launch has its drain and ramp -- and the other stream's workgroups take
# exactly those CUs: 759 -> 742 ms at C2, 245.5 -> 231.0 ms at 90x120, 163.8 -> 152.7 ms at 64x96 (alternating runs on one box;
# round 2 measured it neutral: the direct kernels' many small tiles left no such tails).  The stream has the device's lowest
# priority (the data-gradient chain is the critical path: 745.5 vs 750.2 ms at the default priority).
# Not below 2^14 pixels per launch: at 31x56 bs 4 and 45x80 bs 2 it is neutral on average and bimodal from run to run
# (79.3 ... 86.8 ms against 82.2 / 82.7 ms on one stream; what it gained there in an earlier state, 87.9 -> 82.6 ms, was the
# overlap of slab reductions that have since become 3x shorter).  Not in the bf16 mode (31x56: 66.3 -> 83.6 ms, the bf16
# pixel-reduction kernel and the convolutions then fight for LDS), not in bf16x6 (neutral, 910.5 vs 908.5 ms at C2), and not
# while a HIP graph is captured.  `tools/side_stream_matrix.sh` is the measurement.
# Protocol: the side stream waits for the launch stream before every weight-gradient launch (its operands were just produced
# there); the operand tensors are kept referenced until the join (small problems) or handed to the caching allocator with
# record_stream (large ones: keeping every window's operands until the end of an 8-window backward would not fit); the join
# -- launch stream waits for the side stream -- is an autograd-engine callback at the end of the backward pass that armed it,
# i.e. before anything (optimizer, GradAllReducer.finish) reads a .grad.  Only for gradients that go straight into leaf
# parameters' .grad (sink route): a gradient handed back to autograd stays on the launch stream.
# Accumulation order into a .grad = issue order on the one side stream = backward's order: deterministic as before, and
# bit-identical to the single-stream run (tests/test_gpu_r3.py, tests/test_gpu_r4.py).
WGRAD_SIDE = os.environ.get("BMC_WGRAD_STREAM", "auto")          # "0" never, "1" always, "auto" (default): fp32 mode, launches of
WGRAD_SIDE_MIN_PIXELS = int(os.environ.get("BMC_WGRAD_STREAM_MIN_PIXELS", 1 << 14))     # at least this many pixels (see above)
# operands of a side-stream launch: kept referenced until the join (<= this many pixels) or handed to the caching allocator with
# record_stream (>: the allocator then defers the reuse of every such buffer to the side stream's progress, so peak memory
# follows the side stream's lag).  (Until round 4 this was BMC_WGRAD_STREAM_MAX_PIXELS, which once meant "no side stream
# above this size"; the old name is refused rather than silently reinterpreted.)
if "BMC_WGRAD_STREAM_MAX_PIXELS" in os.environ:
    import warnings
    warnings.warn("BMC_WGRAD_STREAM_MAX_PIXELS is no longer read: the side stream has no upper size limit; "
                  "BMC_WGRAD_KEEP_MAX_PIXELS sets the keep / record_stream threshold of its operands")
WGRAD_KEEP_MAX_PIXELS = int(os.environ.get("BMC_WGRAD_KEEP_MAX_PIXELS", 1 << 17))


class _SideState:
    __slots__ = ("stream", "raw", "event", "keep", "armed", "side", "task", "dev")

    def __init__(self, dev):
        self.dev = dev
        # the device's lowest stream priority: the data-gradient chain (critical path) gets the CUs first, the weight gradients
        # what it leaves (C2: 745.5 vs 750.2 ms at the default priority, alternating runs; small frames: no difference)
        h = C.c_void_p()
        lib.call(lib._stream_low, "bmc_stream_create_low_priority", C.byref(h))
        self.stream = torch.cuda.ExternalStream(h.value, device=dev)
        self.raw = self.stream.cuda_stream
        self.event = torch.cuda.Event()
        self.keep, self.armed, self.side, self.task = [], False, False, -1

    def follow_main(self):
        """Everything queued on the launch stream so far happens before what the side stream is given next."""
        self.event.record(torch.cuda.current_stream(self.dev))
        self.stream.wait_event(self.event)

    def join(self):
        # (the device is named: as an autograd-engine callback this may run on a thread whose current device is not the
        #  model's, and the wait must go onto the model's launch stream)
        if self.side:
            torch.cuda.current_stream(self.dev).wait_stream(self.stream)
        self.keep.clear()
        self.armed = False


_SIDE = {}


def wgrad_join():
    """The launch stream waits for every weight gradient still running on the side stream; idempotent, free when nothing is
    pending.  The backward pass that armed the side stream joins it by itself when it ends -- but a backward pass that RAISED
    never gets there, and if the caller catches the exception the side stream may still be adding into .grad tensors (or into
    blocks zero_grad(set_to_none=True) has meanwhile returned to the allocator).  Called from every place that is about to
    read or free a .grad: the optimizers' step (a global pre-step hook registered below), GradAllReducer.finish(), the models'
    forward, sink_group outside a backward pass.  Code that catches a backward exception and touches .grad by other means
    (clip_grad_norm_, zero_grad followed by raw allocations) calls it first."""
    cur = torch._C._current_graph_task_id()
    if cur < 0 and _MERGE:
        # weight-gradient uses still queued outside a backward pass: the pass that queued them raised (a pass that ends flushes
        # its own queues).  Its gradients are undefined anyway; launching them NOW would add them to whatever .grad holds after
        # the caller's zero_grad -- they are dropped
        _MERGE.clear()
    if cur < 0:
        _FLUSH_QUEUED.clear()
    for st in _SIDE.values():
        if st.armed and (cur < 0 or st.task != cur):      # (a forward recomputed INSIDE the pass that armed it: nothing to join)
            st.join()


def _join_before_step(*_):
    wgrad_join()


from torch.optim.optimizer import register_optimizer_step_pre_hook as _reg_pre_step  # noqa: E402
_reg_pre_step(_join_before_step)


class _SideCtx:
    """Launches issued inside go to the side stream: NOT by switching torch's current stream (two Python context switches per
    launch cost more host time than the overlap buys at these sizes) but by overriding the raw stream handle our launches
    take; torch-side allocations stay on the launch stream's pool and are handed over with _on_side()."""
    __slots__ = ("st",)

    def __init__(self, st):
        self.st = st

    def __enter__(self):
        global _STREAM_OVERRIDE, _STREAM_OVERRIDE_TS
        self.st.follow_main()
        _STREAM_OVERRIDE, _STREAM_OVERRIDE_TS = self.st.raw, self.st.stream
        return self.st

    def __exit__(self, *exc):
        global _STREAM_OVERRIDE, _STREAM_OVERRIDE_TS
        _STREAM_OVERRIDE = _STREAM_OVERRIDE_TS = None
        return False


class _NoCtx:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NOCTX = _NoCtx()



def _side_arm(npx):
    """The side-stream decision of the running backward pass (made by its FIRST weight-gradient use, launched or queued): ->
    the device's _SideState, or None outside a backward pass / with the stream switched off."""
    if WGRAD_SIDE == "0" or torch._C._current_graph_task_id() < 0:
        return None
    dev = _cur_dev()
    st = _SIDE.get(dev)
    if st is None:
        st = _SIDE[dev] = _SideState(dev)
    task = torch._C._current_graph_task_id()
    if st.armed and st.task != task:
        # the backward pass that armed the join never ran its callback (it raised): join now, or this pass would queue no join
        # of its own and the optimizer could read a .grad the side stream is still adding to
        st.join()
    if not st.armed:
        st.task = task
        # the FIRST weight-gradient use of a backward pass decides for the whole pass: one parameter's gradient is
        # accumulated by launches of different batch sizes (conv_fs: B and 3B), and its read-modify-writes must not be
        # split over two streams
        # (not while a HIP graph is being captured: replaying the two-stream graph serialises badly -- 31x56: 174.9 ms against
        #  82.3 ms eager and 88 ms for the one-stream graph; "1" forces it)
        st.side = WGRAD_SIDE == "1" or (MATH == 0 and npx >= WGRAD_SIDE_MIN_PIXELS and not torch.cuda.is_current_stream_capturing())
        # (end-of-pass callbacks run in the order they were queued: what is left in the merge queues leaves BEFORE the join)
        _queue_flush(task)
        torch.autograd.Variable._execution_engine.queue_callback(st.join)
        st.armed = True
    return st


def wgrad_side(npx, params, keep=()):
    """Context for one weight-gradient launch (+ its reduction): the side stream when the mode has it (above), all its
    destinations are sink parameters and an autograd backward pass is running (the join hangs on its end), else nothing.
    keep: the operand tensors of the launch (referenced until the join)."""
    st = _side_arm(npx)
    if st is None or not st.side:
        return _NOCTX
    ps = [p for p in params if p is not None]
    if not ps or not all(is_sink(p) for p in ps):
        return _NOCTX
    if npx <= WGRAD_KEEP_MAX_PIXELS:
        st.keep.extend(keep)
    else:                       # large operands: not kept until the join -- the allocator defers their reuse to the side stream's progress
        for t in keep:
            t.record_stream(st.stream)
    return _SideCtx(st)


def _flat_params(*ps):
    out = []
    for p in ps:
        if isinstance(p, (tuple, list)):
            out.extend(p)
        elif p is not None:
            out.append(p)
    return out


def reduce_wgrad(slabs, nsplit, G, taps, Cout, spec, dev, bias_slabs, w_param, b_param, w_shape):
    """Sum the pixel-reduction GEMM's slabs into the weight (+ bias) gradient.  -> (dw, db) for autograd; entries are
    None where the gradient went straight into a leaf parameter's .grad (see above).  w_param / b_param may be tuples of
    G parameters (one per weight group of a grouped launch: v1 / v2, conv_hp / conv_hn)."""
    want_b = bias_slabs is not None
    full = spec.covers_all
    if isinstance(w_param, (tuple, list)):
        ps = list(w_param) + (list(b_param) if want_b else [])
        sg = sink_group(ps, full) if 1 < G <= 4 and len(w_param) == G else None
        if sg is not None:
            grads, acc = sg
            PA = C.c_void_p * G
            dwp = PA(*[g.data_ptr() for g in grads[:G]])
            dbp = PA(*[g.data_ptr() for g in grads[G:]]) if want_b else None
            lib.call(lib._red_wg, "bmc_pgemm_reduce_weight_groups", slabs.data_ptr(), nsplit, G, taps, Cout, spec.kpad,
                     spec.kmap(dev).data_ptr(), spec.cin, dwp, acc, bias_slabs.data_ptr() if want_b else None, dbp, _stream())
            return None, None
        w_param = b_param = None
    sg = sink_group([w_param, b_param] if want_b else [w_param], full) if G == 1 and w_param is not None else None
    if sg is not None:
        (gw, *rest), acc = sg
        gb = rest[0] if want_b else None
        lib.call(lib._red_w, "bmc_pgemm_reduce_weight", slabs.data_ptr(), nsplit, 1, taps, Cout, spec.kpad,
                 spec.kmap(dev).data_ptr(), spec.cin, gw.data_ptr(), acc, bias_slabs.data_ptr() if want_b else None,
                 gb.data_ptr() if want_b else None, _stream())
        return None, None
    dw = (torch.empty if full else torch.zeros)(G * Cout * spec.cin * taps, device=dev, dtype=torch.float32)
    db = torch.empty((G, Cout), device=dev, dtype=torch.float32) if want_b else None
    lib.call(lib._red_w, "bmc_pgemm_reduce_weight", slabs.data_ptr(), nsplit, G, taps, Cout, spec.kpad,
             spec.kmap(dev).data_ptr(), spec.cin, dw.data_ptr(), 0, bias_slabs.data_ptr() if want_b else None,
             db.data_ptr() if want_b else None, _stream())
    return dw.view(w_shape), ((db[0] if G == 1 else db) if want_b else None)


# --------------------------------------------------------------------------
# convolution (3x3 / 1x1, multi-source, grouped weights)
# --------------------------------------------------------------------------
class ConvMeta:
    __slots__ = ("spec", "views", "B", "relu", "G", "res", "cache", "taps", "out", "ngp", "rule")

    def __init__(self, spec, views, B, relu, G, res, cache, taps, out=None, ngp=0, rule=None):
        self.spec, self.views, self.B, self.relu, self.G, self.res, self.cache, self.taps, self.out, self.ngp, self.rule = \
            spec, views, B, relu, G, res, cache, taps, out, ngp, rule


class ConvFn(torch.autograd.Function):
    """y = epi(conv(cat(sources)) + bias [+ residual]) -- F.conv2d + torch.cat + F.relu + residual add of
    models/submodules.py:31-35,63-64,75 and models/BMCNet.py:64-82; with G > 1 weights are per batch-group
    (torch.bmm(softmax, v) of models/submodules.py:72-73 is the G = B, 1x1 case)."""

    @staticmethod
    def forward(ctx, meta: ConvMeta, weight, bias, res_t, *rest):
        # rest = the source tensors, then (meta.ngp of them) the parameters behind a stacked per-group weight: G weights,
        # then G biases -- `weight` / `bias` are then detached stacks of them (ops.stacked) and the gradients go to these
        src_ts, gps = (rest[:len(rest) - meta.ngp], rest[len(rest) - meta.ngp:]) if meta.ngp else (rest, ())
        for t in src_ts:
            _need_gpu(t)
        t0 = src_ts[0]
        _, H, W, _ = t0.shape
        G, taps, B = meta.G, meta.taps, meta.B
        w4 = weight.detach().reshape(G, -1, meta.spec.cin, taps)
        Cout = w4.shape[1]
        ck = weight if meta.cache else None
        wn = wino_ok(B, H, W, Cout, taps, fwd=True, stride=max([t.shape[3] for t in src_ts] + [res_t.shape[3] if res_t is not None else 0]),
                     rule=meta.rule if meta.rule is not None else bias)
        wp = _packed_weight(w4.contiguous(), meta.spec, ck, wino=wn)
        if meta.out is not None:       # write into a batch range of a preallocated buffer (see OutSlot)
            out = meta.out.t[meta.out.b0:meta.out.b0 + B]
            assert out.shape == (B, H, W, Cout) and out.is_contiguous()
        else:
            out = torch.empty((B, H, W, Cout), device=t0.device, dtype=torch.float32)
        srcs = [_src(t.detach(), *v, B) for t, v in zip(src_ts, meta.views)]
        res = None
        if res_t is not None:
            res = _src(res_t.detach(), 0, Cout, meta.res[0], meta.res[1], 0, B)
        cp = coutpad(Cout)
        conv_raw(srcs, wp, meta.spec.kpad * taps * cp, bias.detach() if bias is not None else None, Cout,
                 out.data_ptr(), H * W * Cout, Cout, B, H, W, Cout, taps, relu=meta.relu, residual=res,
                 bpg=B // G, flops=2.0 * B * H * W * Cout * taps * meta.spec.kreal, wino=wn)
        ctx.meta = meta
        ctx.params = (weight, bias)  # the objects the caller passed (leaf parameters take their gradients directly)
        if meta.ngp:
            ctx.params = (tuple(gps[:G]), tuple(gps[G:]) if len(gps) > G else None)
        ctx.w_owner = ck            # identity of the parameter object (saved_tensors may hand back a new wrapper)
        ctx.has_bias = bias is not None
        ctx.has_res = res_t is not None
        ctx.window = current_window()
        ctx.save_for_backward(weight, out if meta.relu else None, *src_ts)
        return out

    @staticmethod
    def backward(ctx, dy):
        meta = ctx.meta
        weight, out, *src_ts = ctx.saved_tensors
        G, taps, B, spec = meta.G, meta.taps, meta.B, meta.spec
        dy = dy.contiguous()
        g = relu_bwd(dy, out) if meta.relu else dy
        _, H, W, Cout = g.shape
        dev = g.device
        w4 = weight.detach().reshape(G, Cout, spec.cin, taps).contiguous()
        ck = ctx.w_owner
        need = ctx.needs_input_grad
        ngp = meta.ngp
        nsrc = len(src_ts)
        if ngp:      # (weight, bias) are detached stacks: what asks for the gradients are the parameters behind them
            need = list(need)
            need[1] = any(need[4 + nsrc:4 + nsrc + G])
            need[2] = any(need[4 + nsrc + G:])
        dw = db = dres = None
        bias_done = False
        # ---- weight gradient: pixel-reduction GEMM  dW[co][k][tap] = sum_px g[px][co] * x[px+tap][k]
        if need[1]:
            srcs = [_src(t, *v, B) for t, v in zip(src_ts, meta.views)]
            a_src = _src(g, 0, Cout, 0, None, 0, B)
            wb = ctx.has_bias and need[2]
            wp_, bp_ = ctx.params
            v0 = meta.views[0]
            sp = None
            if (WINO and WINO_WGRAD and MATH == 0 and taps == 9 and G == 1 and Cout == 128 and nsrc > 1 and not ngp
                    and is_sink(wp_) and (not wb or is_sink(bp_)) and a_src.pix_stride == 128
                    and all(t.shape[3] == 128 for t, n in zip(src_ts, spec.nch) if n == 128)):
                sp = split_wgrad(spec, meta.views)
            if sp is not None:
                # multi-source convolution on leaf parameters: the 128-channel sources through the Winograd kernel (each its own
                # column window of the weight gradient), the narrow ones through one pixel-reduction launch; all of them add
                # into the same .grad (the first one zero-fills it: none defines every column)
                big, rest, sub = sp
                for n, (i, k0) in enumerate(big):
                    wgrad_wino(a_src, srcs[i], B, H, W, spec, dev, wp_, bp_ if (wb and n == 0) else None, weight.shape,
                               want_bias=wb and n == 0, k0=k0, full=False, keep=(g, src_ts[i]), window=ctx.window)
                if rest:
                    with wgrad_side(B * H * W, [wp_], (g, *src_ts)):
                        r_pg = pgemm_raw(a_src, [srcs[i] for i in rest], B, H, W, taps, B, Cout, sub.kpad, dev,
                                         flops=2.0 * B * H * W * Cout * taps * sub.kreal)
                        reduce_wgrad(r_pg[0], r_pg[1], 1, taps, Cout, sub, dev, None, wp_, None, weight.shape)
                dw = db = None
            elif wino_wgrad_ok(a_src, srcs, spec, taps, Cout, G) and not isinstance(wp_, (tuple, list)):
                dw, db = wgrad_wino(a_src, srcs[0], B, H, W, spec, dev, wp_, bp_ if wb else None, weight.shape, want_bias=wb,
                                    keep=(g, src_ts[0]), window=ctx.window)
            elif (ngp and 1 < G <= 4 and len(wp_) == G and wino_wgrad_ok(a_src, srcs, spec, taps, Cout, 1) and v0[2] == 0
                  and v0[3] is None):
                # grouped launch over separate parameters (conv_hp / conv_hn): one Winograd weight gradient per group, on the
                # group's batch window of both operands
                bpg = B // G
                outs = []
                for gi in range(G):
                    a_g = _src(g, 0, Cout, 0, None, gi * bpg, bpg)
                    x_g = _src(src_ts[0], v0[0], v0[1], 0, None, v0[4] + gi * bpg, bpg)
                    outs.append(wgrad_wino(a_g, x_g, bpg, H, W, spec, dev, wp_[gi], bp_[gi] if wb else None, wp_[gi].shape, want_bias=wb,
                                           keep=(g, src_ts[0]), window=ctx.window))
                dw = None if all(o[0] is None for o in outs) else torch.stack([
                    o[0] if o[0] is not None else torch.zeros_like(wp_[i]) for i, o in enumerate(outs)])
                db = None if (not wb or all(o[1] is None for o in outs)) else torch.stack([
                    o[1] if o[1] is not None else torch.zeros_like(bp_[i]) for i, o in enumerate(outs)])
            else:
                with wgrad_side(B * H * W, _flat_params(wp_, bp_ if wb else None), (g, *src_ts)):
                    r_pg = pgemm_raw(a_src, srcs, B, H, W, taps, B // G, Cout, spec.kpad, dev,
                                     flops=2.0 * B * H * W * Cout * taps * spec.kreal, want_bias=wb)
                    slabs, nsplit = r_pg[0], r_pg[1]
                    dw, db = reduce_wgrad(slabs, nsplit, G, taps, Cout, spec, dev, r_pg[3] if wb else None, wp_,
                                          bp_ if wb else None, weight.shape)
            bias_done = wb
        if ctx.has_bias and need[2] and not bias_done:
            bpg = B // G
            parts = [colsum(g.data_ptr() + 4 * gi * bpg * H * W * Cout, bpg * H * W, Cout, Cout, dev) for gi in range(G)]
            db = parts[0] if G == 1 else torch.stack(parts)
        if ctx.has_res and need[3]:
            shift, mod = meta.res
            dres = g
            if mod is not None and mod < B:
                dres = group_sum(g, B // mod)
            elif shift:
                dres = torch.roll(g, shifts=shift, dims=0)
        # ---- data gradients: same conv kernel, transposed + mirrored weights, one launch per source.
        # Sources that are disjoint windows of ONE tensor (the batch halves of a twin tensor) share one gradient tensor:
        # each launch writes its window, the tensor is handed to autograd once -- no zero fill, no add.
        shared_dx = {}
        simple = lambda v: v[2] == 0 and v[3] is None
        for i, (t, v) in enumerate(zip(src_ts, meta.views)):
            if need[4 + i] and simple(v):
                shared_dx.setdefault(id(t), []).append(i)
        for key, idxs in list(shared_dx.items()):
            t = src_ts[idxs[0]]
            Bt, _, _, Ct = t.shape
            wins = [(meta.views[i][4], meta.views[i][0], meta.views[i][1]) for i in idxs]            # (b0, c0, nch)
            apart = lambda p, q: p[0] + B <= q[0] or q[0] + B <= p[0] or p[1] + p[2] <= q[1] or q[1] + q[2] <= p[1]
            disjoint = all(apart(wins[j], wins[k]) for j in range(len(wins)) for k in range(j + 1, len(wins)))
            inside = all(w[0] + B <= Bt and w[1] + w[2] <= Ct for w in wins)
            if len(idxs) >= 2 and G == 1 and disjoint and inside and sum(B * w[2] for w in wins) == Bt * Ct:
                shared_dx[key] = (idxs, torch.empty_like(t))
            else:
                del shared_dx[key]
        dsrcs = []
        for i, (t, v) in enumerate(zip(src_ts, meta.views)):
            if not need[4 + i]:
                dsrcs.append(None)
                continue
            grp = shared_dx.get(id(t)) if simple(v) else None
            if grp is not None:
                c0, nch, shift, mod, b0 = v
                Bt, _, _, Ct = t.shape
                dxs_ = grp[1]
                wn = wino_ok(B, H, W, nch, taps, stride=Ct)
                wt = _packed_weight_t(w4, spec, i, ck, wino=wn)
                conv_raw([_src(g, 0, Cout, 0, None, 0, B)], wt, round_up(Cout, CK) * taps * coutpad(nch), None, 0,
                         dxs_.data_ptr() + 4 * (b0 * H * W * Ct + c0), H * W * Ct, Ct, B, H, W, nch, taps, bpg=B,
                         flops=2.0 * B * H * W * spec.real_nch[i] * taps * Cout, wino=wn)
                dsrcs.append(dxs_ if i == grp[0][0] else None)
                continue
            c0, nch, shift, mod, b0 = v
            Bt, _, _, Ct = t.shape
            c16 = round_up(Cout, CK)
            nkpad = coutpad(nch)
            gs, nb, gshift, gmod = g, B, 0, None
            post = None
            if G > 1 and ((mod is not None and mod < B) or shift):
                # grouped weights + remapped operand: launch over the full batch, fold the batch map afterwards
                tmp = torch.empty((B, H, W, nch), device=dev, dtype=torch.float32)
                wn = wino_ok(B, H, W, nch, taps)
                wt = _packed_weight_t(w4, spec, i, ck, wino=wn)
                conv_raw([_src(g, 0, Cout, 0, None, 0, B)], wt, c16 * taps * nkpad, None, 0, tmp.data_ptr(), H * W * nch,
                         nch, B, H, W, nch, taps, bpg=B // G, flops=2.0 * B * H * W * spec.real_nch[i] * taps * Cout, wino=wn)
                if mod is not None and mod < B:
                    tmp = tmp.view(B // mod, mod, H, W, nch).sum(0)
                    if shift:
                        tmp = torch.roll(tmp, shifts=shift, dims=0)
                elif shift:
                    tmp = torch.roll(tmp, shifts=shift, dims=0)
                nbt = tmp.shape[0]
                if c0 == 0 and nch == Ct and b0 == 0 and nbt == Bt:
                    dx = tmp
                else:
                    dx = torch.zeros_like(t)
                    dx[b0:b0 + nbt, :, :, c0:c0 + nch] = tmp
                dsrcs.append(dx)
                continue
            if mod is not None and mod < B:          # operand shared by several launch batches: sum first (linearity)
                assert shift == 0
                gs, nb = group_sum(g, B // mod), mod
            elif shift:                               # operand read with a batch rotation: rotate back
                assert mod == B
                gshift, gmod = (mod - shift) % mod, mod
            full = (c0 == 0 and nch == Ct and b0 == 0 and nb == Bt)
            dx = torch.empty_like(t) if full else torch.zeros_like(t)
            gsrc = _src(gs, 0, Cout, gshift, gmod, 0, nb)
            if Cout % CK:
                raise RuntimeError("bmc_hip: conv output channels must be a multiple of 16 for the data gradient")
            wn = wino_ok(nb, H, W, nch, taps, stride=Ct)
            wt = _packed_weight_t(w4, spec, i, ck, wino=wn)
            conv_raw([gsrc], wt, c16 * taps * nkpad, None, 0, dx.data_ptr() + 4 * (b0 * H * W * Ct + c0), H * W * Ct, Ct,
                     nb, H, W, nch, taps, bpg=nb // G, flops=2.0 * nb * H * W * spec.real_nch[i] * taps * Cout, wino=wn)
            dsrcs.append(dx)
        if ngp:      # gradients of the stacked weights' owners: None when they went straight into .grad, else the stack's slices
            wps, bps = ctx.params
            gw = [None] * G if dw is None else [dw[i].reshape(wps[i].shape) for i in range(G)]
            gb = [] if bps is None else ([None] * G if db is None else [db[i] for i in range(G)])
            return (None, None, None, dres, *dsrcs, *gw, *gb)
        return (None, dw, db, dres, *dsrcs)


def conv(views: Sequence[View], weight, bias, spec: ConvSpec, *, B=None, relu=False, residual=None, G=1,
         cache=True, taps=None, out=None, rule=None):
    """views: operands (in packed-K order of `spec`); weight [Cout,Cin,kh,kw] (G == 1) or [G,Cout,Cin(,1,1)];
    out: optional OutSlot -- the result is written into (and returned as a view of) a batch range of its buffer.
    rule: for a bias-free 3x3 launch whose residual carries a bias, that bias (the exact-zero rule above wino_ok)."""
    B = views[0].t.shape[0] if B is None else B
    if taps is None:
        taps = weight.shape[-1] * weight.shape[-2] if weight.dim() >= 4 else 1
    res_t, res_meta = None, None
    if residual is not None:
        res_t, res_meta = residual.t, (residual.shift, residual.mod)
    meta = ConvMeta(spec, [v.meta() for v in views], B, relu, G, res_meta, cache, taps, out, rule=rule)
    return ConvFn.apply(meta, weight, bias, res_t, *[v.t for v in views])


def conv_groups(views: Sequence[View], weights, biases, spec: ConvSpec, *, B=None, relu=False, residual=None, out=None):
    """One launch over G = len(weights) batch groups, group g convolving with the PARAMETERS weights[g] / biases[g]
    (equal shapes; models/BMCNet.py:79-80: conv_hp / conv_hn): the stacked operands are cached per parameter version
    (`stacked`), and the weight / bias gradients of the launch go to the parameters themselves -- straight into their .grad
    where they are leaves (bmc_pgemm_reduce_weight_groups), through autograd otherwise."""
    G = len(weights)
    B = views[0].t.shape[0] if B is None else B
    w0 = weights[0]
    taps = w0.shape[-1] * w0.shape[-2]
    wst = stacked(tuple(weights), lambda: torch.stack([w.detach() for w in weights]), "stack")
    bst = stacked(tuple(biases), lambda: torch.stack([b.detach() for b in biases]), "stack") if biases is not None else None
    res_t, res_meta = None, None
    if residual is not None:
        res_t, res_meta = residual.t, (residual.shift, residual.mod)
    gps = tuple(weights) + (tuple(biases) if biases is not None else ())
    meta = ConvMeta(spec, [v.meta() for v in views], B, relu, G, res_meta, True, taps, out, ngp=len(gps),
                    rule=tuple(biases) if biases is not None else None)
    return ConvFn.apply(meta, wst, bst, res_t, *[v.t for v in views], *gps)


# --------------------------------------------------------------------------
# fused residual block (models/submodules.py:17-35): both ReLU-backward and the skip-path gradient add live in
# convolution epilogues, so the backward is exactly 2 data-gradient + 2 weight-gradient launches (+ bias sums)
# --------------------------------------------------------------------------
def _wgrad_plain(g, x, spec, w_param, b_param, taps, window=None):
    """-> (dW, db) for autograd (None where accumulated into the leaf parameter's .grad): weight gradient and bias
    gradient (column sums of g) from one pgemm launch."""
    B, H, W, Cout = g.shape
    dev = g.device
    a_src, x_src = _src(g, 0, Cout, 0, None, 0, B), _src(x, 0, x.shape[3], 0, None, 0, B)
    if wino_wgrad_ok(a_src, [x_src], spec, taps, Cout, 1):
        return wgrad_wino(a_src, x_src, B, H, W, spec, dev, w_param, b_param, w_param.shape, keep=(g, x), window=window)
    return wgrad_pgemm(a_src, [x_src], B, H, W, taps, Cout, spec, dev, w_param, b_param, w_param.shape, keep=(g, x), window=window)


class GradPair:
    """Where the consumers of the two halves of an un-stacked tensor (bie.Unstack2Fn) put their input gradients: the halves
    of ONE lazily allocated buffer, so that Unstack2Fn.backward returns it as it is instead of concatenating two tensors
    (41 full-size copies per C2 step).  The forward tags the two output views (`_bmc_gslot = (pair, half)`); a consumer that
    knows the protocol (ResBlockFn, BIETwinFn) writes into grad_slot(...) -- any other consumer, or a view with several
    consumers, simply produces an ordinary gradient and the concatenation happens as before."""
    __slots__ = ("n", "shape", "buf", "taken")

    def __init__(self, n, shape):
        self.n, self.shape, self.buf, self.taken = n, tuple(shape), None, [False, False]

    def half(self, i, like):
        """Each half is handed out ONCE (a view with two protocol-aware consumers, or a second backward through a retained
        graph, gets an ordinary tensor: two writers must never share a destination)."""
        if self.taken[i]:
            return torch.empty_like(like)
        self.taken[i] = True
        if self.buf is None:
            self.buf = torch.empty(self.shape, device=like.device, dtype=like.dtype)
        return self.buf[i * self.n:(i + 1) * self.n]


def grad_slot(tag, like):
    """tag: what the forward found on its input (`getattr(x, "_bmc_gslot", None)`); -> the tensor to write dx into."""
    if tag is None:
        return torch.empty_like(like)
    pair, i = tag
    t = pair.half(i, like)
    return t if t.shape == like.shape else torch.empty_like(like)


class OutSlot:
    """Where a Function should put its result: batches [b0, b0 + B) of a preallocated buffer (kept out of autograd's
    sight on purpose -- see bie.Stack2Fn)."""
    __slots__ = ("t", "b0")

    def __init__(self, t, b0):
        self.t, self.b0 = t, b0


class ResBlockFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, spec, out=None):
        _need_gpu(x)
        B, H, W, Cn = x.shape
        taps = w1.shape[-1] * w1.shape[-2]
        cp = coutpad(Cn)
        xs = _src(x.detach(), 0, Cn, 0, None, 0, B)
        wn1, wn2 = wino_ok(B, H, W, Cn, taps, fwd=True, rule=b1), wino_ok(B, H, W, Cn, taps, fwd=True, rule=b2)
        wp1 = _packed_weight(w1.detach().reshape(1, Cn, Cn, taps), spec, w1, wino=wn1)
        wp2 = _packed_weight(w2.detach().reshape(1, Cn, Cn, taps), spec, w2, wino=wn2)
        fl = 2.0 * B * H * W * Cn * taps * Cn
        t = torch.empty_like(x)
        conv_raw([xs], wp1, spec.kpad * taps * cp, b1.detach(), Cn, t.data_ptr(), H * W * Cn, Cn, B, H, W, Cn, taps, relu=True,
                 flops=fl, wino=wn1)
        y = torch.empty_like(x) if out is None else out.t[out.b0:out.b0 + B]
        conv_raw([_src(t, 0, Cn, 0, None, 0, B)], wp2, spec.kpad * taps * cp, b2.detach(), Cn, y.data_ptr(), H * W * Cn, Cn, B, H,
                 W, Cn, taps, residual=xs, flops=fl, wino=wn2)
        ctx.save_for_backward(x, t, w1, w2)
        ctx.spec, ctx.taps = spec, taps
        ctx.owners = (w1, w2)
        ctx.params = (w1, b1, w2, b2)
        ctx.gslot = getattr(x, "_bmc_gslot", None)
        ctx.window = current_window()
        return y

    @staticmethod
    def backward(ctx, g):
        x, t, w1, w2 = ctx.saved_tensors
        spec, taps = ctx.spec, ctx.taps
        g = g.contiguous()
        B, H, W, Cn = g.shape
        dev = g.device
        need = ctx.needs_input_grad
        fl = 2.0 * B * H * W * Cn * taps * Cn
        gs = _src(g, 0, Cn, 0, None, 0, B)
        nkpad, c16 = coutpad(Cn), round_up(Cn, CK)
        p_w1, p_b1, p_w2, p_b2 = ctx.params
        dw2, db2 = _wgrad_plain(g, t, spec, p_w2, p_b2, taps, ctx.window) if (need[3] or need[4]) else (None, None)
        # d(pre-activation of conv1) = ReLU'(t) * conv2^T(g): mask epilogue
        wn = wino_ok(B, H, W, Cn, taps)
        w2t = _packed_weight_t(w2.detach().reshape(1, Cn, Cn, taps), spec, 0, ctx.owners[1], wino=wn)
        dt = torch.empty_like(g)
        conv_raw([gs], w2t, c16 * taps * nkpad, None, 0, dt.data_ptr(), H * W * Cn, Cn, B, H, W, Cn, taps,
                 mask=_src(t, 0, Cn, 0, None, 0, B), flops=fl, wino=wn)
        dw1, db1 = _wgrad_plain(dt, x, spec, p_w1, p_b1, taps, ctx.window) if (need[1] or need[2]) else (None, None)
        dx = None
        if need[0]:   # dx = conv1^T(dt) + g (skip path): residual epilogue
            w1t = _packed_weight_t(w1.detach().reshape(1, Cn, Cn, taps), spec, 0, ctx.owners[0], wino=wn)
            dx = grad_slot(ctx.gslot, g)
            conv_raw([_src(dt, 0, Cn, 0, None, 0, B)], w1t, c16 * taps * nkpad, None, 0, dx.data_ptr(), H * W * Cn, Cn, B, H, W,
                     Cn, taps, residual=gs, flops=fl, wino=wn)
        return dx, dw1, db1, dw2, db2, None, None


def res_block(x, w1, b1, w2, b2, spec, out=None):
    return ResBlockFn.apply(x, w1, b1, w2, b2, spec, out)


# --------------------------------------------------------------------------
# LayerNorm2d (models/submodules.py:127-166)
# --------------------------------------------------------------------------
class LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        _need_gpu(x)
        x = x.contiguous()
        Cn = x.shape[-1]
        npix = x.numel() // Cn
        y = torch.empty_like(x)
        stats = torch.empty(npix * 2, device=x.device, dtype=torch.float32)
        lib.call(lib._ln_fwd, "bmc_layernorm_fwd", x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), npix, Cn, eps,
                 y.data_ptr(), stats.data_ptr(), _stream())
        ctx.save_for_backward(x, stats, gamma)
        ctx.params = (gamma, beta)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, stats, gamma = ctx.saved_tensors
        dy = dy.contiguous()
        Cn = x.shape[-1]
        npix = x.numel() // Cn
        dx = torch.empty_like(x)
        ws = torch.empty(2 * 1024 * Cn, device=x.device, dtype=torch.float32)
        sg = sink_group(list(ctx.params))
        if sg is not None:
            (dg, db), acc = sg
            lib.call(lib._ln_bwd, "bmc_layernorm_bwd", dy.data_ptr(), x.data_ptr(), stats.data_ptr(), gamma.data_ptr(), npix,
                     Cn, dx.data_ptr(), ws.data_ptr(), dg.data_ptr(), db.data_ptr(), acc, _stream())
            return dx, None, None, None
        dg = torch.empty(Cn, device=x.device, dtype=torch.float32)
        db = torch.empty(Cn, device=x.device, dtype=torch.float32)
        lib.call(lib._ln_bwd, "bmc_layernorm_bwd", dy.data_ptr(), x.data_ptr(), stats.data_ptr(), gamma.data_ptr(), npix,
                 Cn, dx.data_ptr(), ws.data_ptr(), dg.data_ptr(), db.data_ptr(), 0, _stream())
        return dx, dg, db, None


def layer_norm(x, gamma, beta, eps=1e-6):
    return LayerNormFn.apply(x, gamma, beta, eps)


# --------------------------------------------------------------------------
# channel Gram matrix + softmax (models/submodules.py:69-73)
# --------------------------------------------------------------------------
_ID_SPECS = {}


def _dense_spec(n):
    s = _ID_SPECS.get(n)
    if s is None:
        s = ConvSpec.dense(n)
        _ID_SPECS[n] = s
    return s


class GramFn(torch.autograd.Function):
    """att[b] = scale * sum_px c[b,px,:]^T v[b,px,:]  ([C,C] per sample) -- torch.bmm(center, v) * scale."""

    @staticmethod
    def forward(ctx, c, v, scale):
        _need_gpu(c)
        B, H, W, Cn = c.shape
        a_src = _src(c.detach(), 0, Cn, 0, None, 0, B)
        slabs, nsplit, G = pgemm_raw(a_src, [_src(v.detach(), 0, Cn, 0, None, 0, B)], B, H, W, 1, 1, Cn, Cn, c.device,
                                     flops=2.0 * B * H * W * Cn * Cn)
        att = torch.empty((B, Cn, Cn), device=c.device, dtype=torch.float32)
        lib.call(lib._red_p, "bmc_pgemm_reduce_plain", slabs.data_ptr(), nsplit, G, Cn, Cn, scale, att.data_ptr(),
                 _stream())
        ctx.save_for_backward(c, v)
        ctx.scale = scale
        return att

    @staticmethod
    def backward(ctx, datt):
        c, v = ctx.saved_tensors
        B, H, W, Cn = c.shape
        spec = _dense_spec(Cn)
        ds = (datt * ctx.scale).contiguous()
        dc = dv = None
        cp = coutpad(Cn)
        if ctx.needs_input_grad[0]:   # dc[px,i] = sum_j ds[i,j] v[px,j]
            wp = _packed_weight(ds.view(B, Cn, Cn, 1), spec, None)
            dc = torch.empty_like(c)
            conv_raw([_src(v, 0, Cn, 0, None, 0, B)], wp, spec.kpad * cp, None, 0, dc.data_ptr(), H * W * Cn, Cn, B, H, W,
                     Cn, 1, bpg=1, flops=2.0 * B * H * W * Cn * Cn)
        if ctx.needs_input_grad[1]:   # dv[px,j] = sum_i ds[i,j] c[px,i]
            wp = _packed_weight(ds.transpose(1, 2).contiguous().view(B, Cn, Cn, 1), spec, None)
            dv = torch.empty_like(v)
            conv_raw([_src(c, 0, Cn, 0, None, 0, B)], wp, spec.kpad * cp, None, 0, dv.data_ptr(), H * W * Cn, Cn, B, H, W,
                     Cn, 1, bpg=1, flops=2.0 * B * H * W * Cn * Cn)
        return dc, dv, None


def gram(c, v, scale):
    return GramFn.apply(c, v, scale)


class SoftmaxFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a):
        _need_gpu(a)
        a = a.contiguous()
        p = torch.empty_like(a)
        Cn = a.shape[-1]
        lib.call(lib._sm_fwd, "bmc_softmax_fwd", a.data_ptr(), a.numel() // Cn, Cn, p.data_ptr(), _stream())
        ctx.save_for_backward(p)
        return p

    @staticmethod
    def backward(ctx, dp):
        (p,) = ctx.saved_tensors
        dp = dp.contiguous()
        Cn = p.shape[-1]
        da = torch.empty_like(p)
        lib.call(lib._sm_bwd, "bmc_softmax_bwd", p.data_ptr(), dp.data_ptr(), p.numel() // Cn, Cn, 1.0, da.data_ptr(),
                 _stream())
        return da


def softmax_rows(a):
    return SoftmaxFn.apply(a)


def attn_apply(p, v, residual: Optional[View] = None):
    """out[b,px,i] = sum_j p[b,i,j] v[b,px,j] (+ residual) -- torch.bmm(softmax, v^T) of models/submodules.py:72-73
    as a 1x1 convolution with one weight matrix per sample."""
    B, _, _, Cn = v.shape
    return conv([View(v)], p, None, _dense_spec(Cn), B=B, residual=residual, G=B, cache=False, taps=1)


# --------------------------------------------------------------------------
# window head / tail (models/BMCNet.py:106-119, models/submodules.py:80-92)
# --------------------------------------------------------------------------
def pack_inputs(x, repeat=3):
    """x [B,2,T,H,W] (any strides) -> xin12 NHWC [2B,H,W,16]: batches [0,B) carry the positive-polarity frames
    [f1,f1,f1,f2,f2,f2,0..], batches [B,2B) the negative ones; no gradient (network inputs)."""
    _need_gpu(x)
    B, _, _, H, W = x.shape
    xin = torch.empty((2 * B, H, W, CK), device=x.device, dtype=torch.float32)
    sb, sc, st, sy, sx = x.stride()
    lib.call(lib._pack_in, "bmc_pack_inputs", x.data_ptr(), sb, sc, st, sy, sx, B, H, W, repeat, xin.data_ptr(),
             xin.data_ptr() + 4 * B * H * W * CK, _stream())
    return xin


def _unshuffle(hr, r, split=1):
    B, Cc, HH, WW = hr.shape
    H, W = HH // r, WW // r
    lr = torch.empty((split * B, H, W, Cc * r * r // split), device=hr.device, dtype=torch.float32)
    lib.call(lib._unshuffle, "bmc_unshuffle_to_nhwc", hr.data_ptr(), B, Cc, H, W, r, lr.data_ptr(), split, _stream())
    return lr


def _shuffle(lr, r, base=None, split=1):
    Bs, H, W, cg = lr.shape
    B, CC = Bs // split, cg * split
    Cc = CC // (r * r)
    hr = torch.empty((B, Cc, H * r, W * r), device=lr.device, dtype=torch.float32)
    if base is None:
        lib.call(lib._shuffle, "bmc_shuffle_to_hr", lr.data_ptr(), B, Cc, H, W, r, None, 0, 0, 0, 0, hr.data_ptr(), split,
                 _stream())
    else:
        sb, sc, sy, sx = base.stride()
        lib.call(lib._shuffle, "bmc_shuffle_to_hr", lr.data_ptr(), B, Cc, H, W, r, base.data_ptr(), sb, sc, sy, sx,
                 hr.data_ptr(), split, _stream())
    return hr


class UnshuffleFn(torch.autograd.Function):
    """HR NCHW [B,C,rH,rW] -> LR NHWC [B,H,W,C r r] (PixelUnShuffle, models/submodules.py:80-92); split = S stores it as S
    batch-stacked channel groups [S B,H,W,C r r / S] (what the input-fusion convolutions read: no cat in between)."""

    @staticmethod
    def forward(ctx, hr, r, split):
        _need_gpu(hr)
        ctx.r, ctx.split = r, split
        return _unshuffle(hr.contiguous(), r, split)

    @staticmethod
    def backward(ctx, dlr):
        return _shuffle(dlr.contiguous(), ctx.r, None, ctx.split), None, None


class HeadFn(torch.autograd.Function):
    """pred = pixel_shuffle(x_o, r) + bilinear_up(base, r)  (models/BMCNet.py:119); base carries no gradient."""

    @staticmethod
    def forward(ctx, xo, base, r):
        _need_gpu(xo)
        ctx.r = r
        return _shuffle(xo.contiguous(), r, base)

    @staticmethod
    def backward(ctx, dpred):
        return _unshuffle(dpred.contiguous(), ctx.r), None, None


class HeadMseFn(torch.autograd.Function):
    """(pred, mse) = head + nn.MSELoss in one pass (bmc_head_mse_fwd); the backward folds the loss gradient
    2 (pred - gt) / numel into the pixel-unshuffle of the gradient arriving from the next window (bmc_head_mse_bwd)."""

    @staticmethod
    def forward(ctx, xo, base, gt, r):
        _need_gpu(xo)
        xo = xo.contiguous()
        B, H, W, CC = xo.shape
        Cc = CC // (r * r)
        if tuple(gt.shape) != (B, Cc, H * r, W * r) or not gt.is_cuda or gt.dtype != torch.float32:
            raise RuntimeError("head_mse: ground truth must be a float32 GPU tensor of the prediction's shape")
        if gt.stride()[1:] != (H * r * W * r, W * r, 1):
            gt = gt.contiguous()
        pred = torch.empty((B, Cc, H * r, W * r), device=xo.device, dtype=torch.float32)
        ws = torch.empty(2048, device=xo.device, dtype=torch.float32)
        loss = torch.empty((), device=xo.device, dtype=torch.float32)
        sb, sc, sy, sx = base.stride()
        lib.call(lib._head_mse_fwd, "bmc_head_mse_fwd", xo.data_ptr(), B, Cc, H, W, r, base.data_ptr(), sb, sc, sy, sx,
                 gt.data_ptr(), gt.stride(0), pred.data_ptr(), ws.data_ptr(), loss.data_ptr(), _stream())
        ctx.r = r
        ctx.dims = (B, Cc, H, W)
        ctx.save_for_backward(pred, gt)
        return pred, loss

    @staticmethod
    def backward(ctx, dpred, dloss):
        pred, gt = ctx.saved_tensors
        B, Cc, H, W = ctx.dims
        r = ctx.r
        if dpred is not None:
            dpred = dpred.contiguous()
        if dloss is not None:
            dloss = dloss.contiguous()
        dlr = torch.empty((B, H, W, Cc * r * r), device=pred.device, dtype=torch.float32)
        lib.call(lib._head_mse_bwd, "bmc_head_mse_bwd", dpred.data_ptr() if dpred is not None else None, pred.data_ptr(),
                 gt.data_ptr(), gt.stride(0), dloss.data_ptr() if dloss is not None else None, B, Cc, H, W, r,
                 dlr.data_ptr(), _stream())
        return dlr, None, None, None


def head_mse(xo, base, gt, r):
    return HeadMseFn.apply(xo, base, gt, r)


def pixel_unshuffle_nhwc(hr, r, split=1):
    return UnshuffleFn.apply(hr, r, split)


class StackViewsFn(torch.autograd.Function):
    """NCHW-shaped, channels-last-strided tensors that already lie back to back in one allocation (the recurrent states
    the previous window's convolutions wrote into one buffer) -> the NHWC tensor [sum B,H,W,C] over the same memory: no
    copy.  Backward hands every part its slice of the gradient as a view."""

    @staticmethod
    def forward(ctx, *parts):
        p0 = parts[0]
        B, Cc, H, W = p0.shape
        ctx.nb = [p.shape[0] for p in parts]
        out = torch.empty(0, device=p0.device, dtype=p0.dtype)
        out.set_(p0.untyped_storage(), p0.storage_offset(), (sum(ctx.nb), H, W, Cc), (H * W * Cc, W * Cc, Cc, 1))
        return out

    @staticmethod
    def backward(ctx, g):
        outs, b = [], 0
        for n in ctx.nb:
            outs.append(g[b:b + n].permute(0, 3, 1, 2))
            b += n
        return tuple(outs)


def stack_states(parts):
    """[B,C,H,W] tensors -> NHWC [sum B,H,W,C]; free when they are adjacent channels-last views of one buffer (see
    StackViewsFn), one copy (torch.cat) otherwise (first window: the caller's separate zero tensors)."""
    p0 = parts[0]
    B, Cc, H, W = p0.shape
    want = (H * W * Cc, 1, W * Cc, Cc)
    ok = all(p.is_cuda and p.dtype == torch.float32 and tuple(p.shape[1:]) == (Cc, H, W) and p.stride() == want for p in parts)
    if ok:
        off = p0.storage_offset()
        for p in parts:
            if p.untyped_storage().data_ptr() != p0.untyped_storage().data_ptr() or p.storage_offset() != off:
                ok = False
                break
            off += p.shape[0] * H * W * Cc
    if ok:
        return StackViewsFn.apply(*parts)
    return torch.cat([p.permute(0, 2, 3, 1).contiguous() for p in parts], 0)


def head(xo, base, r):
    return HeadFn.apply(xo, base, r)


class BicubicResizeFn(torch.autograd.Function):
    """F.interpolate(x, size=size, mode='bicubic', align_corners=False) on NCHW tensors (train.py:227-231)."""

    @staticmethod
    def forward(ctx, x, Ho, Wo):
        _need_gpu(x)
        x = x.contiguous()
        B, Cc, H, W = x.shape
        y = torch.empty((B, Cc, Ho, Wo), device=x.device, dtype=torch.float32)
        lib.call(lib._bicubic_fwd, "bmc_bicubic_resize_fwd", x.data_ptr(), B * Cc, H, W, Ho, Wo, y.data_ptr(), _stream())
        ctx.shape = (B, Cc, H, W)
        return y

    @staticmethod
    def backward(ctx, gy):
        B, Cc, H, W = ctx.shape
        gy = gy.contiguous()
        gx = torch.empty((B, Cc, H, W), device=gy.device, dtype=torch.float32)
        lib.call(lib._bicubic_bwd, "bmc_bicubic_resize_bwd", gy.data_ptr(), B * Cc, H, W, gy.shape[2], gy.shape[3],
                 gx.data_ptr(), _stream())
        return gx, None, None


def bicubic_resize(x, size):
    """x [B,C,H,W] -> [B,C,size[0],size[1]]; identity (no launch) when the sizes already agree."""
    if tuple(x.shape[-2:]) == (int(size[0]), int(size[1])):
        return x
    return BicubicResizeFn.apply(x, int(size[0]), int(size[1]))


# --------------------------------------------------------------------------
# event -> count image (dataloader/encodings.py:290-305)
# --------------------------------------------------------------------------
def events_to_channels_batched(xs, ys, ps, offsets, H, W, mutate=True):
    """xs/ys/ps: fp32 device vectors; offsets: int64 device vector [nframes+1] -> [nframes,2,H,W]."""
    _need_gpu(xs)
    nframes = offsets.numel() - 1
    out = torch.empty((nframes, 2, H, W), device=xs.device, dtype=torch.float32)
    ws = _binned_ws(xs.numel(), nframes, H, W, xs.device)
    if ws is not None:       # large frames: LDS-privatised count images, no scattered global float atomics (csrc/scatter.hip)
        lib.call(lib._events_binned, "bmc_events_to_channels_binned", xs.data_ptr(), ys.data_ptr(), ps.data_ptr(),
                 offsets.data_ptr(), xs.numel(), nframes, H, W, out.data_ptr(), int(mutate), ws.data_ptr(), ws.numel() * 8,
                 _stream())
        return out
    lib.call(lib._events, "bmc_events_to_channels", xs.data_ptr(), ys.data_ptr(), ps.data_ptr(), offsets.data_ptr(),
             nframes, H, W, out.data_ptr(), int(mutate), _stream())
    return out


BINNED_MIN_PIXELS = int(os.environ.get("BMC_BINNED_MIN_PIXELS", 1 << 17))


def _binned_ws(nevents, nframes, H, W, device):
    """Workspace of the binned event encoders, or None where the plain atomic kernel is the better fit (small frames: a
    count image that fits a few LDS bands anyway; the HR ground-truth frames of the step, 720x960, take the binned path)."""
    if H * W < BINNED_MIN_PIXELS or nevents == 0 or nframes == 0 or W > 16384:
        return None
    nbytes = lib._events_ws(nevents, nframes, H, W)
    if nbytes < 0:
        return None
    return torch.empty((nbytes + 7) // 8, device=device, dtype=torch.int64)


def encode_raw_events(xs_i16, ys_i16, ps_f64, offsets, flips, H, W):
    """Raw dataset columns (int16, int16, float64 device vectors) + per-frame flip flags (uint8 or None)
    -> [nframes,2,H,W] count images; see bmc_encode_raw_events."""
    for t, dt in ((xs_i16, torch.int16), (ys_i16, torch.int16), (ps_f64, torch.float64), (offsets, torch.int64)):
        if not t.is_cuda or t.dtype != dt or not t.is_contiguous():
            raise RuntimeError("encode_raw_events: expected contiguous %s tensors on the GPU (no CPU fallback)" % dt)
    nframes = offsets.numel() - 1
    out = torch.empty((nframes, 2, H, W), device=xs_i16.device, dtype=torch.float32)
    fp = None
    if flips is not None:
        if flips.dtype != torch.uint8 or not flips.is_cuda or flips.numel() != nframes:
            raise RuntimeError("encode_raw_events: flips must be a uint8 GPU vector with one entry per frame")
        fp = flips.data_ptr()
    ws = _binned_ws(xs_i16.numel(), nframes, H, W, xs_i16.device)
    if ws is not None:
        lib.call(lib._enc_raw_binned, "bmc_encode_raw_events_binned", xs_i16.data_ptr(), ys_i16.data_ptr(), ps_f64.data_ptr(),
                 offsets.data_ptr(), fp, xs_i16.numel(), nframes, H, W, out.data_ptr(), ws.data_ptr(), ws.numel() * 8, _stream())
        return out
    lib.call(lib._enc_raw, "bmc_encode_raw_events", xs_i16.data_ptr(), ys_i16.data_ptr(), ps_f64.data_ptr(),
             offsets.data_ptr(), fp, nframes, H, W, out.data_ptr(), _stream())
    return out


def events_to_voxel_batched(xs, ys, ts, ps, offsets, bins, H, W, mutate=True):
    """fp32 device vectors + int64 frame offsets -> [nframes, bins, H, W] temporal-bilinear voxel grids."""
    _need_gpu(xs)
    nframes = offsets.numel() - 1
    n = xs.numel()
    out = torch.empty((nframes, bins, H, W), device=xs.device, dtype=torch.float32)
    ws = torch.empty(2 * nframes * (H * W + 1) + n, device=xs.device, dtype=torch.int32)
    lib.call(lib._voxel, "bmc_events_to_voxel", xs.data_ptr(), ys.data_ptr(), ts.data_ptr(), ps.data_ptr(),
             offsets.data_ptr(), n, nframes, bins, H, W, out.data_ptr(), int(mutate), ws.data_ptr(), _stream())
    return out


def _bin_bounds(ts, bins):
    """float32 bin bounds with the reference's own expressions (dataloader/encodings.py:171-176,224-229), on the device."""
    dt = ts[-1] - ts[0] + 1e-6
    delta_t = dt / bins
    bi = torch.arange(bins, device=ts.device, dtype=torch.float32)
    tstart = ts[0] + delta_t * bi
    return tstart, tstart + delta_t


def events_to_stack_polarity(xs, ys, ts, ps, bins, H, W, mutate=True):
    """fp32 device vectors of ONE event window -> [2, bins, H, W] polarity-split event stack (events_to_stack_polarity)."""
    _need_gpu(xs)
    tstart, tend = _bin_bounds(ts, bins)
    out = torch.empty((2, bins, H, W), device=xs.device, dtype=torch.float32)
    ranges = torch.empty(2 * bins, device=xs.device, dtype=torch.int32)
    lib.call(lib._stack_pol, "bmc_events_to_stack_polarity", xs.data_ptr(), ys.data_ptr(), ts.data_ptr(), ps.data_ptr(),
             ts.numel(), tstart.data_ptr(), tend.data_ptr(), bins, H, W, out.data_ptr(), ranges.data_ptr(), int(mutate),
             _stream())
    return out


def events_to_mask(xs, ys, ps, H, W, mutate=True):
    """fp32 device vectors -> [H, W] event mask (events_to_mask); xs / ys / ps lose their out-of-range entries in place."""
    _need_gpu(xs)
    out = torch.empty((H, W), device=xs.device, dtype=torch.float32)
    ws = torch.empty(H * W, device=xs.device, dtype=torch.int32)
    lib.call(lib._mask, "bmc_events_to_mask", xs.data_ptr(), ys.data_ptr(), ps.data_ptr(), xs.numel(), H, W, out.data_ptr(),
             ws.data_ptr(), int(mutate), _stream())
    return out


def events_to_stack(xs, ys, ts, ps, bins, H, W, mutate=True):
    """fp32 device vectors of ONE event window -> [bins, H, W] event stack (events_to_stack_no_polarity).  The bin
    bounds are evaluated here with the reference's own float32 expressions (dataloader/encodings.py:224-229) on the
    device; search + scatter + in-place masking run in libbmc_hip.so."""
    _need_gpu(xs)
    n = ts.numel()
    dt = ts[-1] - ts[0] + 1e-6
    delta_t = dt / bins
    bi = torch.arange(bins, device=ts.device, dtype=torch.float32)
    tstart = ts[0] + delta_t * bi
    tend = tstart + delta_t
    out = torch.empty((bins, H, W), device=xs.device, dtype=torch.float32)
    ranges = torch.empty(2 * bins, device=xs.device, dtype=torch.int32)
    lib.call(lib._stack, "bmc_events_to_stack", xs.data_ptr(), ys.data_ptr(), ts.data_ptr(), ps.data_ptr(), n,
             tstart.data_ptr(), tend.data_ptr(), bins, H, W, out.data_ptr(), ranges.data_ptr(), int(mutate), _stream())
    return out
