import torch
torch.manual_seed(0)
def split3(x):
    h=x.to(torch.bfloat16).float(); r=x-h
    m=r.to(torch.bfloat16).float(); r2=r-m
    l=r2.to(torch.bfloat16).float()
    return h,m,l
def split3_trunc(x):
    def tr(v): return (v.view(torch.int32) & -65536).view(torch.float32)
    h=tr(x); r=x-h; m=tr(r); r2=r-m; l=tr(r2); return h,m,l
def mm6(a,w,split=split3):
    a1,a2,a3=split(a); w1,w2,w3=split(w)
    # products exact in fp32 accumulate (emulate with float64 accumulate then round?) -> use fp32 matmul of bf16-valued fp32 tensors: products exact, sums fp32
    f=lambda x,y:(x.double()@y.double())
    return (f(a1,w1)+f(a1,w2)+f(a2,w1)+f(a1,w3)+f(a2,w2)+f(a3,w1)).float()
def mm3(a,w):
    a1,a2,_=split3(a); w1,w2,_=split3(w)
    f=lambda x,y:(x.double()@y.double())
    return (f(a1,w1)+f(a1,w2)+f(a2,w1)).float()
N=4096
a=torch.randn(N,32).relu()*torch.rand(N,1)*3; w=torch.randn(32,32)*0.2
ref=a.double()@w.double()
def err(x): return ((x.double()-ref).abs().max()/ref.abs().max()).item(), ((x.double()-ref).abs()/ (a.abs().double()@w.abs().double())).max().item()
print("fp32 matmul", err(a@w))
print("bf16x3 6 products", err(mm6(a,w)))
print("bf16x3 trunc 6 products", err(mm6(a,w,split3_trunc)))
print("bf16x2 3 products", err(mm3(a,w)))
h,m,l=split3(a); print("split exact?", (h+m+l-a).abs().max().item())
h,m,l=split3_trunc(a); print("trunc split exact?", (h+m+l-a).abs().max().item())
