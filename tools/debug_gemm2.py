import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package()
L = pkg.load_library()
from lstm_rnn_amd import binding as B
ctx = C.c_void_p(); B.check(L.cn_ctx_create(0, 0, None, C.byref(ctx)))
M, N, K = 32, 32, 32
np.set_printoptions(linewidth=250)
for k0 in (0, 1, 2, 4, 5, 8, 17):
    A = np.zeros((M, K), np.float32); A[:, k0] = 1
    Bm = np.tile(np.arange(K, dtype=np.float32), (N, 1))
    Cc = np.zeros((M, N), np.float32)
    B.check(L.cn_dbg_gemm_nt(ctx, A.ctypes.data, Bm.ctypes.data, Cc.ctypes.data, M, N, K, None, 2), ctx)
    print("k0", k0, "C uniq", np.unique(Cc))
A = np.tile(np.arange(M, dtype=np.float32)[:, None], (1, K)); Bm = np.zeros((N, K), np.float32); Bm[:, 3] = 1
Cc = np.zeros((M, N), np.float32)
B.check(L.cn_dbg_gemm_nt(ctx, A.ctypes.data, Bm.ctypes.data, Cc.ctypes.data, M, N, K, None, 2), ctx)
print("rows:", Cc[:, 0])
A = np.zeros((M, K), np.float32); A[:, 3] = 1; Bm = np.tile(np.arange(N, dtype=np.float32)[:, None], (1, K))
B.check(L.cn_dbg_gemm_nt(ctx, A.ctypes.data, Bm.ctypes.data, Cc.ctypes.data, M, N, K, None, 2), ctx)
print("cols:", Cc[0, :])
