import torch, time
dev=torch.device('cuda:0')
q=torch.empty(621780,32,device=dev)
src=torch.randn(621780,32,device=dev)
def t(fn,n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e)/n*1e3
print('zero_ 80MB us', t(lambda: q.zero_()))
print('copy_ 80MB us', t(lambda: q.copy_(src)))
print('fill us', t(lambda: q.fill_(1.0)))
