"""Streaming single-GPU inference with the reference's semantics (infer_BMCNet.py:20-103):
the recurrent state (h, h_p, h_n, previous HR prediction) is created once and carried across calls, every call
runs one window under no_grad, and the per-window latency is measured with events on the launch stream exactly
where the reference puts its `starter/ender` pair (infer_BMCNet.py:54,66-68)."""
import torch


class StreamingSR:
    """graph=True: from the third window on, a window is ONE HIP-graph replay (the ~300 kernel launches of a window are
    captured once, input and recurrent state live in static buffers) -- at small sensor sizes the eager window is bound by
    the host's launch rate, not by the GPU."""

    def __init__(self, model, n_c=128, scale=4, plain=False, graph=False):
        self.model = model.eval()
        self.n_c, self.scale, self.plain = n_c, scale, plain
        self.use_graph = graph
        self.reset()

    def reset(self):
        self.state = None
        self.times_ms = []
        self._graph = self._x_static = self._out_static = None
        self._calls = 0

    def _capture(self, x):
        """Capture `state <- model(x_static, state, False)` with the state in static buffers."""
        self._x_static = x.clone()
        self._state_static = [t.clone() for t in self.state]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                      # warm-up on a side stream, as the capture protocol asks
            self.model(self._x_static, *self._state_static, False)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = self.model(self._x_static, *self._state_static, False)
            for dst, src in zip(self._state_static, out):
                dst.copy_(src)
        self._graph = g

    @torch.no_grad()
    def step(self, x, timed=True):
        """x [B,2,T>=2,H,W] on the GPU (inp_cnt.transpose(1,2) of the reference) -> HR prediction [B,2,sH,sW]."""
        B, _, _, H, W = x.shape
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
            self.state = tuple(out)
        elif self.use_graph and self._calls >= 3:
            if self._graph is None:
                self._capture(x)
            self._x_static.copy_(x)
            self._graph.replay()
            self.state = tuple(self._state_static)
            out = self.state
        else:
            out = self.model(x, *self.state, False)
            self.state = tuple(out)
        if timed:
            end.record()
            end.synchronize()
            self.times_ms.append(start.elapsed_time(end))
        return out[-1]

    @staticmethod
    @torch.no_grad()
    def esr_mse(pred, gt):
        """The evaluation metric of infer_BMCNet.py:76-84: bicubic-resize the prediction to the ground truth's size when
        they differ (:77-78; EventZoom: 124x224 vs 124x222), then the mean squared error."""
        from bmc_hip import ops
        return torch.nn.functional.mse_loss(ops.bicubic_resize(pred, gt.shape[-2:]), gt)

    def latency_ms(self, skip=1):
        """Mean per-window latency (the reference's `time` metric), ignoring the first `skip` windows."""
        t = self.times_ms[skip:] or self.times_ms
        return sum(t) / max(len(t), 1)
