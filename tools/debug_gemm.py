import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package()
L = pkg.load_library()
from lstm_rnn_amd import binding as B
rng = np.random.RandomState(0)
for prec in (0, 1):
    ctx = C.c_void_p(); B.check(L.cn_ctx_create(0, prec, None, C.byref(ctx)))
    for (M, N, K) in [(60, 128, 32), (200, 256, 64), (1000, 1024, 256), (130, 192, 96)]:
        A = rng.randn(M, K).astype(np.float32); Bm = rng.randn(N, K).astype(np.float32); bias = rng.randn(N).astype(np.float32)
        Cc = np.zeros((M, N), np.float32)
        B.check(L.cn_dbg_gemm_nt(ctx, A.ctypes.data, Bm.ctypes.data, Cc.ctypes.data, M, N, K, bias.ctypes.data, 2), ctx)
        ref = A.astype(np.float64) @ Bm.astype(np.float64).T + bias
        print("prec", prec, "NT", M, N, K, "max err", np.abs(Cc - ref).max(), "ref max", np.abs(ref).max())
    for (M, N, K) in [(128, 32, 60), (256, 64, 1000), (1024, 256, 5000), (96, 160, 333)]:
        A = rng.randn(K, M).astype(np.float32); Bm = rng.randn(K, N).astype(np.float32)
        Cc = np.zeros((M, N), np.float32)
        B.check(L.cn_dbg_gemm_tn(ctx, A.ctypes.data, Bm.ctypes.data, Cc.ctypes.data, M, N, K), ctx)
        ref = A.astype(np.float64).T @ Bm.astype(np.float64)
        print("prec", prec, "TN", M, N, K, "max err", np.abs(Cc - ref).max(), "ref max", np.abs(ref).max())
    L.cn_ctx_destroy(ctx)
