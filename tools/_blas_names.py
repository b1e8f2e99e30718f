import torch
for (M,N,K,tA,tB) in [(24000,1200,1024,0,0),(640,1200,400,0,0),(400,1200,640,1,0),(640,400,1200,0,1),(24000,400,200,0,1),(24000,200,400,0,0)]:
    A = torch.randn((K,M) if tA else (M,K), device="cuda"); B = torch.randn((N,K) if tB else (K,N), device="cuda")
    for _ in range(3):
        C = torch.matmul(A.t() if tA else A, B.t() if tB else B)
torch.cuda.synchronize()
