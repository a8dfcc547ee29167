import torch
dev=torch.device('cuda')
a=torch.empty(600*64*64*32,device=dev)
b=torch.randn_like(a)
def t(fn,n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/n*1e3
nb=a.numel()*4
us=t(lambda:a.fill_(1.0)); print('fill 315MB: %.1f us %.2f TB/s'%(us,nb/us/1e6))
us=t(lambda:a.copy_(b)); print('copy 315MB->315MB: %.1f us %.2f TB/s (r+w)'%(us,2*nb/us/1e6))
us=t(lambda:b.sum()); print('sum-read 315MB: %.1f us %.2f TB/s'%(us,nb/us/1e6))
