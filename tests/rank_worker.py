"""One rank of a world-size-N run of the REAL training step on the HIP path (a fresh interpreter per rank, started by
tests/test_gpu_r3.py with subprocess -- never a re-exec of a process that touched the GPU):

    python tests/rank_worker.py <out.npz> <backend: gloo|nccl> <device index>      (RANK / WORLD_SIZE / MASTER_* from the env)

models.BMCNet(4,16,1) + train_step.shard_sequences + bmc_hip.parallel.GradAllReducer + train_step.bptt_step -- the replacement
of the reference's DDP scaffolding (train.py:62-83,227-237; dataloader/h5dataloader.py:191-201).  BMC_ACCUM_GRADS in the
environment selects the gradient route (1: the kernels add into .grad and the reducer stages at step time; 0: autograd
accumulates, post-accumulate hooks fill and launch the buckets during backward).  Writes the rank's loss and every
parameter gradient after optimizer.step()'s pre-hook (= the averaged gradients)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.dirname(HERE), os.path.join(os.path.dirname(HERE), "bmcnet-esr_amd")]


def problem(dev):
    import torch
    scale, n_c, n_b, B, L, H, W = 4, 16, 1, int(os.environ.get("BMC_RANK_TEST_B", 4)), 3, 12, 20
    from models.BMCNet import BMCNet
    torch.manual_seed(6)
    m = BMCNet(scale, n_c, n_b).to(dev)
    g = torch.Generator().manual_seed(9)
    inp = torch.poisson(torch.full((B, L, 2, H, W), 0.4), generator=g).to(dev)
    gt = torch.poisson(torch.full((B, L, 2, scale * H, scale * W), 0.4), generator=g).to(dev)
    return m, inp, gt, n_c, scale


def main():
    out, backend, devi = sys.argv[1], sys.argv[2], int(sys.argv[3])
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(devi)
    dev = torch.device("cuda", devi)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from bmc_hip import ops
    from bmc_hip.parallel import GradAllReducer
    from train_step import bptt_step, shard_sequences
    m, inp, gt, n_c, scale = problem(dev)
    opt = torch.optim.SGD(m.parameters(), lr=0.0)
    red = GradAllReducer(m, opt, bucket_mb=0.05)
    launched_from_hooks = []
    launch0 = red._launch
    red._launch = lambda bi: (launched_from_hooks.append(bi), launch0(bi))[1]
    si, sg = shard_sequences(inp, gt, rank, world)
    in_finish = []
    finish0 = red.finish
    red.finish = lambda: (in_finish.append(len(launched_from_hooks)), finish0())[1]     # (the step pre-hook calls red.finish())
    loss, _ = bptt_step(m, opt, si, sg, n_c, scale)
    torch.cuda.synchronize()
    np.savez(out, loss=float(loss), accum=int(ops.ACCUM_PARAM_GRADS), hook_launches=in_finish[0], nbuckets=len(red.buckets),
             side_stream=int(any(st.side for st in ops._SIDE.values())),
             **{"g%03d" % i: (p.grad if p.grad is not None else torch.zeros_like(p)).detach().cpu().numpy()       # (None: a parameter
                for i, p in enumerate(m.parameters())})                                                         #  the loss does not reach)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
