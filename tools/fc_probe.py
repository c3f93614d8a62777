import torch, time
x=torch.randn(16128,128,device='cuda',requires_grad=True); W=torch.randn(128,128,device='cuda',requires_grad=True); b=torch.randn(128,device='cuda',requires_grad=True)
def t(fn,n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter()-t0)/n*1e6
print("F.linear", t(lambda: torch.nn.functional.linear(x,W,b)))
Wt=W.t().contiguous()
print("addmm Wt", t(lambda: torch.addmm(b,x,Wt)))
print("addmm W.t()", t(lambda: torch.addmm(b,x,W.t())))
print("matmul+add", t(lambda: x@W.t()+b))
print("matmul Wt + add", t(lambda: x@Wt+b))
print("mm only Wt", t(lambda: torch.mm(x,Wt)))
print("mm only W.t()", t(lambda: torch.mm(x,W.t())))
