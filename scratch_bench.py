import sys, time, torch
sys.path.insert(0, '.')
from picopose_amd.utils import matching as hm
B,N,C = [int(x) for x in sys.argv[1:4]] if len(sys.argv)>3 else (32,162,768)
dev='cuda'
g=torch.Generator(device=dev).manual_seed(1)
bank=torch.randn(B,N,C,16,16,device=dev,generator=g)
q=torch.randn(B,C,16,16,device=dev,generator=g)
yy,xx=torch.meshgrid(torch.arange(224.0),torch.arange(224.0),indexing='ij')
m=(((yy-111.5)**2+(xx-111.5)**2)<(0.4*224)**2).float()[None].repeat(B,1,1).to(dev)
for mode in ['exact','fast']:
    for _ in range(3): s,st=hm.template_scores(bank,q,m,mode=mode,return_stats=True)
    torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    it=10
    for _ in range(it): s=hm.template_scores(bank,q,m,mode=mode)
    e1.record(); torch.cuda.synchronize()
    ms=e0.elapsed_time(e1)/it
    gb=(B*N*C*256*4+B*C*256*4)/1e9
    print(f"{mode}: {ms:.3f} ms  {gb/ms*1e3:.1f} GB/s  {B/ms*1e3:.1f} crops/s stats={st.tolist()}", flush=True)
