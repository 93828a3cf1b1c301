import sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import test_gpu_training_attention as T
for B, heads, L, qs in [(2,4,256,1.0),(3,4,1024,1.0),(1,2,4096,1.0),(2,4,256,12.0),(2,4,1024,4.0),(1,2,4096,3.0),(1,4,4096,6.0)]:
    g = torch.Generator().manual_seed(1234 + L)
    q = torch.randn(B, heads, L, 16, generator=g) * qs
    k = torch.randn(B, heads, L, 16, generator=g)
    v = torch.randn(B, heads, L, 16, generator=g) * 2 + 0.5
    dout = torch.randn(B, L, heads * 16, generator=g)
    w = T.core_reference(q, k, v, dout)
    o = T.run_core(q, k, v, dout)
    print(L, qs, " ".join(f"{n}:{((a.double()-b).abs().max()/b.abs().max()).item():.2e}" for n, a, b in zip(("out","dq","dk","dv"), o, w)), flush=True)
