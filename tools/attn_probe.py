import sys,os,torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cremage_amd import ops
B,N,C,H=(8,4096,320,8) if len(sys.argv) < 2 else tuple(int(v) for v in sys.argv[1].split(","))
q=torch.randn(B,N,C,device="cuda").to(torch.bfloat16); k=torch.randn(B,N,C,device="cuda").to(torch.bfloat16)
vt=torch.randn(B,C,N,device="cuda").to(torch.bfloat16)
for _ in range(3): ops.attention(q,k,vt,H,N,(C//H)**-0.5)
torch.cuda.synchronize()
e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): ops.attention(q,k,vt,H,N,(C//H)**-0.5)
e1.record(); torch.cuda.synchronize()
print("attn us", e0.elapsed_time(e1)*100)
