import os, sys, torch
sys.path.insert(0, "/root/repo")
from cremage_amd import ops
torch.manual_seed(0)
B, N, heads, d = 2, 256, 8, 40
C = heads * d
q = torch.randn(B, N, C, device="cuda").to(torch.bfloat16); k = torch.randn(B, N, C, device="cuda").to(torch.bfloat16); v = torch.randn(B, N, C, device="cuda").to(torch.bfloat16)
vt = v.transpose(1, 2).contiguous()
o = ops.attention(q, k, vt, heads, N, d ** -0.5).float()
sp = lambda t: t.float().reshape(B, N, heads, d).permute(0, 2, 1, 3)
s = torch.einsum("bhid,bhjd->bhij", sp(q), sp(k)) * d ** -0.5
ref = torch.einsum("bhij,bhjd->bhid", s.softmax(-1), sp(v)).permute(0, 2, 1, 3).reshape(B, N, C)
print("rel", ((o - ref).norm() / ref.norm()).item(), "max", (o - ref).abs().max().item(), "finite", torch.isfinite(o).all().item())
print(o[0, 0, :8], ref[0, 0, :8])
