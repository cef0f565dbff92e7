import torch
dev="cuda:0"
def t(fn,n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1)/n*1e3
for (N,K,F) in [(262144,115,128),(262144,128,128),(262144,128,64),(262144,64,32),(262144,32,16),(65536,147,128),(65536,160,128),(16384,211,128),(16384,224,128),(4096,339,128),(1024,593,128)]:
    a=torch.randn(N,K,device=dev); w=torch.randn(K,F,device=dev); b=torch.randn(F,device=dev)
    us=t(lambda: torch.addmm(b,a,w))
    # strided A (row stride padded to multiple of 32)
    Kp=(K+31)//32*32
    ap=torch.zeros(N,Kp,device=dev); ap[:,:K]=a
    wp=torch.zeros(Kp,F,device=dev); wp[:K]=w
    us2=t(lambda: torch.addmm(b,ap,wp))
    us3=t(lambda: torch.addmm(b,ap[:,:K],w))
    print(N,K,F,"dense %.1f us  padded-K(%d) %.1f us  strided-view %.1f us  (%.1f TF)"%(us,Kp,us2,us3,2*N*K*F/us/1e6))
