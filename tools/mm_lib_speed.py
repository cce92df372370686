"""How fast are the vendor library's fp32 GEMMs on the training step's shapes?  (torch.mm -> hipBLASLt / rocBLAS; a yardstick for
the hand-written k_gemm_nt2 / k_gemm_tn2, which run these at 90-105 TFLOP/s.)"""
import torch, time
torch.backends.cuda.matmul.allow_tf32 = False
dev = torch.device("cuda:0")
shapes = [("gi     x[5120,486] . W^T[486,3072]", 5120, 3072, 486, False),
          ("dX     dG[5120,3072] . W[3072,486]", 5120, 486, 3072, False),
          ("dW_hh  dG^T[3072,5120] . h[5120,1024]", 3072, 1024, 5120, True),
          ("dW_ih  dG^T[3072,5120] . x[5120,486]", 3072, 486, 5120, True),
          ("conv1  [5120,486] . [486,486]", 5120, 486, 486, False),
          ("stacked gi [10240,306] . [306,3072]", 10240, 3072, 306, False)]
for name, M, N, K, ta in shapes:
    a = torch.randn((K, M) if ta else (M, K), device=dev)
    b = torch.randn(K, N, device=dev)
    aa = a.t() if ta else a
    for _ in range(5):
        c = aa @ b
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 50
    for _ in range(n):
        c = aa @ b
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print("%-44s %8.1f us  %6.1f TFLOP/s" % (name, dt * 1e6, 2.0 * M * N * K / dt / 1e12))
