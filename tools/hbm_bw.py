import torch, sys
dev=torch.device("cuda:0")
for mb in (354, 708):
    n=mb*1024*1024//4
    x=torch.randn(n,device=dev); y=torch.empty_like(x)
    for _ in range(3): y.copy_(x)
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): y.copy_(x)
    e1.record(); torch.cuda.synchronize()
    ms=e0.elapsed_time(e1)/20
    print("copy %d MB: %.3f ms = %.2f TB/s (read+write)"%(mb,ms,2*n*4/ms/1e9))
    e0.record()
    for _ in range(20): y.add_(x)
    e1.record(); torch.cuda.synchronize()
    ms=e0.elapsed_time(e1)/20
    print("add_ %d MB: %.3f ms = %.2f TB/s (2 reads + write)"%(mb,ms,3*n*4/ms/1e9))
