import torch, time
a = torch.empty(512*1024*1024//4, device="cuda"); b = torch.empty_like(a)
a.zero_(); 
for _ in range(3): b.copy_(a)
torch.cuda.synchronize(); t0=time.perf_counter()
for _ in range(20): b.copy_(a)
torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/20
print("copy 512 MB: %.1f us, read+write %.2f TB/s" % (dt*1e6, 2*a.numel()*4/dt/1e12))
for _ in range(3): b.fill_(1.0)
torch.cuda.synchronize(); t0=time.perf_counter()
for _ in range(20): b.fill_(1.0)
torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/20
print("fill 512 MB: %.1f us, write %.2f TB/s" % (dt*1e6, a.numel()*4/dt/1e12))
s = torch.zeros((), device="cuda")
for _ in range(3): s = a.sum()
torch.cuda.synchronize(); t0=time.perf_counter()
for _ in range(20): s = a.sum()
torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/20
print("sum 512 MB: %.1f us, read %.2f TB/s" % (dt*1e6, a.numel()*4/dt/1e12))
