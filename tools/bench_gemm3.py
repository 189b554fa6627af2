import torch
M=32768; dev="cuda"
def t(fn, flop, name, n=8):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    print("%-40s %8.3f ms %8.1f TF/s" % (name, ms, flop / ms / 1e9), flush=True)
g = lambda *s: torch.randn(*s, device=dev, dtype=torch.bfloat16)
for N,K in ((12288,4096),(12288,4480),(12288,4608),(12288,4352),(4096,4224),(4096,4352),(22016,4352),(22016,4608),(4096,11136),(4096,11264)):
    x,W=g(M,K),g(N,K); y=torch.empty(M,N,device=dev,dtype=torch.bfloat16)
    t(lambda: torch.mm(x,W.t(),out=y), 2*M*N*K, "fwd NT N=%d K=%d"%(N,K))
# strided lhs (xa view with row stride > K) as used without LoRA
x=g(M,4480); W=g(12288,4480); y=torch.empty(M,12288,device=dev,dtype=torch.bfloat16)
t(lambda: torch.mm(x[:,:4096],W[:,:4096].t(),out=y), 2*M*12288*4096, "fwd NT strided views K=4096 of 4480")
