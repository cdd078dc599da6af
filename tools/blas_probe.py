import torch, time
dev="cuda"
def timeit(fn,n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b)/n*1e-3
for M,N,K in [(21276,2304,768),(21276,3072,768),(21276,768,768),(21276,768,3072),(8192,8192,8192)]:
    x=torch.randn(M,K,device=dev,dtype=torch.bfloat16); w=torch.randn(N,K,device=dev,dtype=torch.bfloat16)
    dy=torch.randn(M,N,device=dev,dtype=torch.bfloat16)
    t=timeit(lambda: x@w.t()); t2=timeit(lambda: dy@w); t3=timeit(lambda: dy.t()@x)
    fl=2.0*M*N*K
    print(f"torch(hipBLASLt/rocBLAS) M{M} N{N} K{K}: fwd {fl/t/1e12:6.0f}  dgrad {fl/t2/1e12:6.0f}  wgrad {fl/t3/1e12:6.0f} TFLOP/s")
