"""What plain torch elementwise kernels reach on this box at res5 sizes ([2048,7,7,2048] bf16): add (2 reads + 1 write), copy, relu_, sum --
the streaming rates (5.9-6.0 TB/s for copy / add on the round-6 boxes) the BatchNorm row-walk kernels are held against."""
import torch, time
x=torch.randn(2048,7,7,2048,device='cuda').to(torch.bfloat16); y=torch.randn_like(x); z=torch.empty_like(x)
def t(fn,n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True); s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e)/n
gb=x.numel()*2/1e9
a=t(lambda: torch.add(x,y,out=z)); print('add 2r1w', round(a*1e3,1),'us', round(3*gb/a*1e3/1e3,2),'TB/s')
a=t(lambda: z.copy_(x)); print('copy 1r1w', round(a*1e3,1),'us', round(2*gb/a*1e3/1e3,2),'TB/s')
a=t(lambda: torch.relu_(z)); print('relu_ 1r1w', round(a*1e3,1),'us', round(2*gb/a*1e3/1e3,2),'TB/s')
a=t(lambda: x.float().sum()); print('sum 1r', round(a*1e3,1),'us')
a=t(lambda: x.sum(dtype=torch.float32)); print('sum 1r', round(a*1e3,1),'us', round(gb/a*1e3/1e3,2),'TB/s')
