"""Streaming single-GPU inference with the reference's semantics (infer_BMCNet.py:20-103):
the recurrent state (h, h_p, h_n, previous HR prediction) is created once and carried across calls, every call
runs one window under no_grad, and the per-window latency is measured with events on the launch stream exactly
where the reference puts its `starter/ender` pair (infer_BMCNet.py:54,66-68)."""
import torch


class StreamingSR:
    """graph=True: from the third window on, a window is ONE HIP-graph replay (the ~300 kernel launches of a window are
    captured once, input and recurrent state live in static buffers) -- at small sensor sizes the eager window is bound by
    the host's launch rate, not by the GPU.

    The tensor step() returns belongs to the caller in both modes: in graph mode it is a copy of the static prediction
    buffer (the next replay overwrites that buffer).  The captured graph is tied to the input shape and to the
    parameter values it was captured with (the packed weight images are baked into it): a different input shape raises,
    a parameter update (load_state_dict, optimizer step, in-place edit -- anything that bumps a parameter's version
    counter) makes the next step() capture afresh; edits through `.data` bypass the counters -- call reset() or
    invalidate() after them.

    state_dtype=torch.bfloat16: the recurrent FEATURE state (the n_c-channel tensors h, h_p, h_n: 3 x n_c x H x W floats per
    sequence, 66 MB at 180x240 -- what a server multiplexing many sequences through one model keeps resident per sequence) is
    carried in bf16 between windows: rounded to nearest-even when a window hands it over, widened back (exactly) when the next
    one reads it.  The previous HR prediction stays fp32 (it is the caller's output).  Contract: the reference's recurrence
    with round_bf16 applied to the three feature states between windows -- oracle/bmc_oracle.py::round_bf16, pinned by
    tests/test_gpu_r5.py::test_streaming_state_in_bf16_vs_oracle_with_state_rounding; the arithmetic inside a window is unchanged."""

    def __init__(self, model, n_c=128, scale=4, plain=False, graph=False, state_dtype=None):
        if state_dtype not in (None, torch.float32, torch.bfloat16):
            raise ValueError("StreamingSR: state_dtype must be None / torch.float32 / torch.bfloat16 (got %r)" % (state_dtype,))
        self.model = model.eval()
        self.n_c, self.scale, self.plain = n_c, scale, plain
        self.use_graph = graph
        self.state_dtype = None if state_dtype is torch.float32 else state_dtype
        self.reset()

    def _pack(self, out):
        """Model outputs (feature states ..., prediction) -> the carried state: features in state_dtype, prediction fp32."""
        if self.state_dtype is None:
            return tuple(out)
        return tuple(t.to(self.state_dtype) for t in out[:-1]) + (out[-1],)

    def _unpack(self, state):
        """The carried state -> what the model reads (fp32; bf16 -> fp32 is exact)."""
        if self.state_dtype is None:
            return tuple(state)
        return tuple(t.float() for t in state[:-1]) + (state[-1],)

    def state_bytes(self):
        """Bytes of recurrent state carried for the current sequence (0 before the first window)."""
        return 0 if self.state is None else sum(t.numel() * t.element_size() for t in self.state)

    def reset(self):
        """Forget the recurrent state, the timings and the captured graph with its static buffers."""
        self.state = None
        self.times_ms = []
        self._calls = 0
        self.invalidate()

    def invalidate(self):
        """Drop the captured graph (the next graph-mode step() captures again); the recurrent state is kept."""
        if getattr(self, "_graph", None) is not None and self.state is not None:
            self.state = tuple(t.clone() for t in self._state_static)      # the state outlives the static buffers
        self._graph = self._x_static = self._state_static = self._stamp = None

    def _weights_stamp(self):
        return tuple((id(p), p._version) for p in self.model.parameters())

    def _capture(self, x):
        """Capture `state <- model(x_static, state, False)` with the state in static buffers."""
        self._x_static = x.clone()
        self._state_static = [t.clone() for t in self.state]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                      # warm-up on a side stream, as the capture protocol asks
            self.model(self._x_static, *self._unpack(self._state_static), False)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = self.model(self._x_static, *self._unpack(self._state_static), False)
            for dst, src in zip(self._state_static, out):
                dst.copy_(src)                              # (a bf16 static buffer: copy_ rounds to nearest-even, as _pack does)
        self._graph = g
        self._stamp = self._weights_stamp()

    @torch.no_grad()
    def step(self, x, timed=True):
        """x [B,2,T>=2,H,W] on the GPU (inp_cnt.transpose(1,2) of the reference; T = 3 with the reference's default
        SEQN, infer_BMCNet.py:147 -- only frames 0 and 1 are read, models/BMCNet.py:106-107) -> HR prediction [B,2,sH,sW]."""
        B, _, _, H, W = x.shape
        if self.state is not None and tuple(self.state[0].shape) != (B, self.n_c, H, W):
            raise RuntimeError("StreamingSR: input %s does not match the carried state %s; call reset() to start a new "
                               "sequence" % (tuple(x.shape), tuple(self.state[0].shape)))
        start = end = None
        if timed:
            start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            start.record()
        self._calls += 1
        if self.state is None:
            z = lambda c: torch.zeros(B, c, H, W, device=x.device)
            if self.plain:
                out = self.model(x, z(self.n_c), z(2 * self.scale ** 2), True)
            else:
                out = self.model(x, z(self.n_c), z(self.n_c), z(self.n_c), z(2 * self.scale ** 2), True)
            self.state = self._pack(out)
            pred = out[-1]
        elif self.use_graph and self._calls >= 3:
            if self._graph is not None and self._stamp != self._weights_stamp():
                self.invalidate()                           # the graph replays the OLD packed weights
            if self._graph is None:
                self._capture(x)
            if tuple(x.shape) != tuple(self._x_static.shape):
                raise RuntimeError("StreamingSR(graph=True): input shape %s differs from the captured %s; call reset()"
                                   % (tuple(x.shape), tuple(self._x_static.shape)))
            self._x_static.copy_(x)
            self._graph.replay()
            self.state = tuple(self._state_static)
            pred = self._state_static[-1].clone()           # the caller's own copy: the next replay rewrites the buffer
        else:
            out = self.model(x, *self._unpack(self.state), False)
            self.state = self._pack(out)
            pred = out[-1]
        if timed:
            end.record()
            end.synchronize()
            self.times_ms.append(start.elapsed_time(end))
        return pred

    @staticmethod
    @torch.no_grad()
    def esr_mse(pred, gt):
        """The evaluation metric of infer_BMCNet.py:76-84: bicubic-resize the prediction to the ground truth's size when
        they differ (:77-78; EventZoom: 124x224 vs 124x222), then the mean squared error."""
        from bmc_hip import ops
        return torch.nn.functional.mse_loss(ops.bicubic_resize(pred, gt.shape[-2:]), gt)

    @staticmethod
    @torch.no_grad()
    def bicubic_mse(inp_cnt, gt, gt_size=None):
        """The baseline metric of infer_BMCNet.py:79,85: the LR count image of the window's middle frame (inp_cnt[:, 1],
        [B,2,H,W]) bicubic-upsampled to gt_sensor_resolution (default: the ground truth's size) against the ground truth."""
        from bmc_hip import ops
        size = tuple(gt.shape[-2:]) if gt_size is None else tuple(gt_size)
        return torch.nn.functional.mse_loss(ops.bicubic_resize(inp_cnt.contiguous(), size), gt)

    def latency_ms(self, skip=1):
        """Mean per-window latency (the reference's `time` metric), ignoring the first `skip` windows."""
        t = self.times_ms[skip:] or self.times_ms
        return sum(t) / max(len(t), 1)
