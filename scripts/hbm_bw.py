import torch
def t(fn,n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s=torch.cuda.Event(True); e=torch.cuda.Event(True); s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e)/n
for mb in (64, 256, 1024, 4096):
    n=mb*1024*1024//4
    a=torch.randn(n,device='cuda'); b=torch.empty_like(a)
    tc=t(lambda: b.copy_(a)); ts=t(lambda: a.sum()); tw=t(lambda: b.fill_(1.0))
    print(f"{mb} MiB: copy {2*n*4/tc/1e9:.2f} TB/s (r+w)  sum(read) {n*4/ts/1e9:.2f} TB/s  fill(write) {n*4/tw/1e9:.2f} TB/s")
