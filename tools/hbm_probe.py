import torch, time
n = 65536*4884
x = torch.empty(n, dtype=torch.float64, device='cuda')
y = torch.empty(n, dtype=torch.float64, device='cuda')
def t(f, it=10):
    f(); torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/it*1e-3
b = n*8
print('fill   %.2f TB/s' % (b/t(lambda: x.fill_(1.0))/1e12))
print('zero   %.2f TB/s' % (b/t(lambda: x.zero_())/1e12))
print('copy   %.2f TB/s (r+w)' % (2*b/t(lambda: y.copy_(x))/1e12))
print('sum    %.2f TB/s' % (b/t(lambda: x.sum())/1e12))
