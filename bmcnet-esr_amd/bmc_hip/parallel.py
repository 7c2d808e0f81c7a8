"""Data-parallel gradient synchronisation for sequence-batch sharding: one process per GPU,
torch.distributed (backend "nccl" = RCCL over xGMI on ROCm; "gloo" in CPU tests).

The reference has only dead DDP scaffolding (train.py:62-83, myutils/utils.py:41-53 with a hard-coded
world size of 1); what a working version of it needs is one sum-all-reduce of the 10.9 MB fp32 gradient
per step.  Every parameter is shared across the 5 blocks and the 8 recurrent windows, so almost all
gradients become final only at the very end of backward; buckets are therefore few and large: a gradient is
copied into its flat bucket by a post-accumulate hook the moment autograd finalises it, a full bucket is
all-reduced on a side stream while backward continues, and the optimizer's pre-step hook joins the side
stream, averages and points .grad at the reduced buffer.  The training loop itself stays
`loss.backward(); optimizer.step()` as in train.py:236-237.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


class GradAllReducer:
    def __init__(self, module: torch.nn.Module, optimizer: torch.optim.Optimizer | None = None, group=None,
                 bucket_mb: float = 4.0):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.params = [p for p in module.parameters() if p.requires_grad]   # parameters() de-duplicates aliases
        dev = self.params[0].device
        # reverse registration order ~ order in which gradients finalise (heads first, input convs last)
        order = list(reversed(self.params))
        self.buckets = []
        cur, cur_bytes = [], 0
        for p in order:
            cur.append(p)
            cur_bytes += p.numel() * 4
            if cur_bytes >= bucket_mb * (1 << 20):
                self.buckets.append(cur); cur, cur_bytes = [], 0
        if cur:
            self.buckets.append(cur)
        self.flat = [torch.zeros(sum(p.numel() for p in b), device=dev, dtype=torch.float32) for b in self.buckets]
        self.slot = {}
        for bi, b in enumerate(self.buckets):
            off = 0
            for p in b:
                self.slot[p] = (bi, off)
                off += p.numel()
        self.pending = [len(b) for b in self.buckets]
        self.works = [None] * len(self.buckets)
        self.seen = set()            # parameters whose hook fired since the last finish()
        self.deferred = False        # a second backward before step() was detected (gradient accumulation)
        self.cuda = dev.type == "cuda"
        self.dev = dev               # named in every current_stream() below: hooks run on autograd's device threads, whose
                                     # current device need not be the model's on a multi-GPU process
        self.side = torch.cuda.Stream(device=dev) if self.cuda else None
        # gloo with device buckets (the world-size-2 test of the HIP path on ONE GPU; RCCL refuses two ranks on one device):
        # the bucket travels through host memory -- independent of whether this gloo build takes device tensors
        self.host_stage = self.cuda and dist.is_initialized() and dist.get_backend(group) == "gloo"
        self._handles = []
        # The parameters (and the optimizer's pre-step hook) hold the reducer through these hooks: it lives as long as the model
        # does, whether or not the caller keeps a reference -- `GradAllReducer(model, opt)` as a bare statement is a complete
        # set-up.  Dropping the caller's reference therefore does NOT end it: detach() does.
        for p in self.params:
            # bmc_hip.ops.is_sink: OUR hook does not need the autograd route (finish() stages sink gradients) -- but somebody
            # else's hook on the parameter (clipping, logging, a second reducer) does: such a parameter keeps the autograd
            # route.  is_sink recognises our hook by its id, not by the number of hooks present.
            h = p.register_post_accumulate_grad_hook(self._on_grad)
            self._handles.append(h)
            ids = getattr(p, "_bmc_sink_hooks", None)
            if ids is None:
                ids = p._bmc_sink_hooks = set()
            ids.add(h.id)
            p._bmc_sink_touched = False
        self._step_hook = optimizer.register_step_pre_hook(lambda *_: self.finish()) if optimizer is not None else None

    def detach(self):
        """Undo the constructor: hooks removed, the parameters no longer marked for the kernels' sink route on this reducer's
        behalf (a model that outlives its reducer must not keep bypassing autograd for hooks registered later).  The ONLY way to
        end a reducer: its hooks keep it alive as long as the model lives (see the constructor), so `del reducer` changes nothing."""
        for p, h in zip(self.params, self._handles):
            getattr(p, "_bmc_sink_hooks", set()).discard(h.id)
            h.remove()
        self._handles = []
        if self._step_hook is not None:
            self._step_hook.remove()
            self._step_hook = None

    # -- called by autograd once per parameter per backward, after all its uses have been accumulated
    def _on_grad(self, p):
        if p in self.seen and not self.deferred:
            # second backward() before optimizer.step(): p.grad now holds the ACCUMULATED local gradient and the
            # buckets already in flight carry stale sums.  Let them land (so nothing is overwritten under a running
            # all-reduce), then stop launching from hooks: finish() re-stages every gradient and reduces once.
            self._join()
            self.deferred = True
        self.seen.add(p)
        if self.deferred:
            return
        if p.grad is None:
            # autograd ran this parameter's AccumulateGrad node with an UNDEFINED gradient (every use of it handed autograd None: the
            # kernels add its gradient straight into .grad -- bmc_hip.ops.sink_group) and none of those adds has been launched yet:
            # merged weight gradients (ops.wgrad_wino / wgrad_pgemm) are still queued and leave at the end of the pass.  finish()
            # stages the bucket from .grad.  (Found by `bench.py --gpus 2` at 31x56, tests/test_gpu_r6.py: round 5's merge queues and
            # the reducer had never met.)
            return
        if getattr(p, "_bmc_sink_touched", False):
            # a kernel has already added into this parameter's .grad during THIS backward (bmc_hip.ops.sink_group) and now
            # autograd accumulates into it too: more sink adds may follow, so the bucket is not complete when its hook
            # count says so.  Leave it to finish().
            return
        bi, off = self.slot[p]
        self.flat[bi][off:off + p.numel()].copy_(p.grad.reshape(-1))
        self.pending[bi] -= 1
        if self.pending[bi] == 0:
            self._launch(bi)

    def _join(self):
        for i, w in enumerate(self.works):
            if w is not None:
                w.wait()
                self.works[i] = None
        if self.cuda:
            torch.cuda.current_stream(self.dev).wait_stream(self.side)

    def _launch(self, bi):
        if not dist.is_initialized():
            return
        if self.host_stage:
            host = self.flat[bi].cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group)
            self.flat[bi].copy_(host)
            return
        if self.cuda:
            self.side.wait_stream(torch.cuda.current_stream(self.dev))
            with torch.cuda.stream(self.side):
                self.works[bi] = dist.all_reduce(self.flat[bi], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        else:
            self.works[bi] = dist.all_reduce(self.flat[bi], op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def finish(self):
        """Join outstanding all-reduces, average, and expose the result as .grad (views of the flat buckets).
        Every rank runs the same graph, so which parameters received a gradient is the same on every rank.
        Buckets whose parameters did not all come through the hooks -- gradient accumulation over several backward()
        calls, or gradients that the kernels accumulated straight into .grad (bmc_hip.ops.ACCUM_PARAM_GRADS: no autograd
        accumulation, hence no hook) -- are staged and reduced here."""
        if self.cuda:
            from . import ops
            ops.wgrad_join()        # weight gradients still being added on the side stream (a backward that raised never joined)
        nb = len(self.buckets)
        active = [any(p.grad is not None for p in self.buckets[bi]) for bi in range(nb)]
        for bi in range(nb):
            if not active[bi]:
                continue
            # a bucket the hooks completed is in flight and final -- unless a kernel added into one of its parameters' .grad
            # behind autograd's back (sink route): such a bucket is re-staged from .grad, whatever its hook count says
            sunk = any(getattr(p, "_bmc_sink_touched", False) for p in self.buckets[bi])
            if not self.deferred and self.pending[bi] == 0 and not sunk:
                continue
            launched = self.pending[bi] == 0 and not self.deferred
            w = self.works[bi]
            if w is not None:
                w.wait(); self.works[bi] = None
            if launched and self.cuda:
                torch.cuda.current_stream(self.dev).wait_stream(self.side)
            for p in self.buckets[bi]:
                o = self.slot[p][1]
                dst = self.flat[bi][o:o + p.numel()]
                if p.grad is None:                         # parameter the loss did not reach
                    dst.zero_()
                elif p.grad.data_ptr() != dst.data_ptr():
                    dst.copy_(p.grad.reshape(-1))
                elif launched and self.world > 1:
                    # .grad IS the bucket (zero_grad(set_to_none=False) keeps last step's views) and an all-reduce of the
                    # half-finished bucket has already summed other ranks' data into it: the local gradient is gone
                    raise RuntimeError("GradAllReducer: a parameter received gradients both through autograd and straight "
                                       "from the kernels while its .grad aliases the reduction bucket; use "
                                       "optimizer.zero_grad(set_to_none=True) or BMC_ACCUM_GRADS=0")
            self._launch(bi)
        self._join()
        for bi in range(nb):
            if not active[bi]:
                continue
            if self.world > 1:
                self.flat[bi].div_(self.world)
            for p in self.buckets[bi]:
                if p.grad is not None:
                    o = self.slot[p][1]
                    p.grad = self.flat[bi][o:o + p.numel()].view_as(p)
        self.pending = [len(b) for b in self.buckets]
        self.works = [None] * nb
        self.seen = set()
        self.deferred = False
        for p in self.params:
            p._bmc_sink_touched = False


def reduce_tensor(t: torch.Tensor, group=None) -> torch.Tensor:
    """Mean of a scalar over ranks for logging (myutils/utils.py:41-53, with the real world size)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return t.detach().clone()
    r = t.detach().clone()
    dist.all_reduce(r, op=dist.ReduceOp.SUM, group=group)
    return r / dist.get_world_size(group)
