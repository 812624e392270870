import torch, time
dev='cuda:0'
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/n*1e3
for M in (32768, 65536, 131072):
    a,b,c=[torch.randn(M,512,device=dev).bfloat16() for _ in range(3)]
    d=torch.empty_like(a)
    us=t(lambda: torch.add(a,b,out=d))
    print(f"M={M}: a+b->d (2R+1W, {3*a.numel()*2/1e6:.0f} MB): {us:.1f} us = {3*a.numel()*2/us/1e6:.2f} TB/s")
    us=t(lambda: d.copy_(a))
    print(f"M={M}: copy (1R+1W): {us:.1f} us = {2*a.numel()*2/us/1e6:.2f} TB/s")
    us=t(lambda: torch.addcmul(a,b,c,out=d))
    print(f"M={M}: addcmul (3R+1W, {4*a.numel()*2/1e6:.0f} MB): {us:.1f} us = {4*a.numel()*2/us/1e6:.2f} TB/s")
