import torch, time
x=torch.empty((65536,256,10,2),device='cuda',dtype=torch.float32)
for fn,name in ((lambda: x.fill_(1.0),'fill'),(lambda: x.zero_(),'zero'),(lambda: torch.rand(x.shape,out=x),'rand')):
    fn(); torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(20): fn()
    torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/20
    print(name, '%.3f ms'%(dt*1e3), '%.2f TB/s'%(x.numel()*4/dt/1e12))
