"""Which kernels does the vendor library pick for the step's fp32 GEMM shapes?  (tools only: a yardstick, never product); the
kernel names carry the macro tile, the MFMA shape and the K depth.  Run with the interpreter itself behind `--` -- never through
a shebang / env hop, which this pool refuses under the profiler:

    cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d <out> -- python3 $GRAFT_REPO_ROOT/tools/vendor_names.py
"""
import torch
torch.backends.cuda.matmul.allow_tf32 = False
dev = torch.device("cuda:0")
for M in (8192, 65536):
    for N, K in ((512, 512), (512, 480), (256, 512), (128, 256)):
        X = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev); b = torch.randn(N, device=dev)
        Y = torch.empty(M, N, device=dev); dY = torch.randn(M, N, device=dev); dX = torch.empty(M, K, device=dev)
        dW = torch.empty(N, K, device=dev)
        for _ in range(5):
            torch.addmm(b, X, W.t(), out=Y)
            torch.mm(dY, W, out=dX)
            torch.mm(dY.t(), X, out=dW)
        torch.cuda.synchronize()
