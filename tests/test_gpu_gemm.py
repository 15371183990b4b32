"""-m gpu: the MFMA GEMM kernels on their own (cn_dbg_gemm_nt / cn_dbg_gemm_tn), against float64 numpy products.
The small shapes run the 128 x 128 / 64 x 64 kernels of cn_gemm.hip, the large ones the 256 x 256 LDS-DMA kernel of
cn_gemm_big.hip (>= 384 tiles, K in whole k-tiles) and the 128 x 128 gradient tiles; M and N are deliberately not
multiples of the tile sizes."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def bf16_round(a):
    """round-to-nearest-even to bf16, returned as float32 (what the operand conversion kernel does)."""
    u = a.astype(np.float32).view(np.uint32)
    r = ((u >> 16) & 1) + 0x7FFF
    return ((u + r) & 0xFFFF0000).astype(np.uint32).view(np.float32)


@pytest.fixture(scope="module")
def lib(pkg):
    from lstm_rnn_amd import binding as B
    return pkg.load_library(), B


@pytest.mark.parametrize("prec", [0, 1, 2])
@pytest.mark.parametrize("shape", [(60, 128, 32), (1000, 1024, 256), (130, 192, 96),
                                   (4200, 6176, 256), (6500, 4128, 512), (25000, 1024, 1024), (9001, 3104, 160), (12500, 2048, 512),
                                   (16500, 4128, 512)])      # 1105 tiles of 256 x 256 at K = 512: the persistent kernel's short-K rule
def test_gemm_nt(lib, prec, shape):
    L, B = lib
    M, N, K = shape
    rng = np.random.RandomState(M + N + K)
    A = rng.randn(M, K).astype(np.float32); Bm = rng.randn(N, K).astype(np.float32); bias = rng.randn(N).astype(np.float32)
    out = np.zeros((M, N), np.float32)
    ctx = C.c_void_p(); B.check(L.cn_ctx_create(0, prec, None, C.byref(ctx)))
    try:
        B.check(L.cn_dbg_gemm_nt(ctx, A.ctypes.data, Bm.ctypes.data, out.ctypes.data, M, N, K, bias.ctypes.data, 2), ctx)
    finally:
        L.cn_ctx_destroy(ctx)
    if prec == 1:
        A, Bm = bf16_round(A), bf16_round(Bm)
    # (float32 matmul of the reference on sub-blocks keeps the test fast; the error budget below covers it)
    ref = A @ Bm.T + bias
    tol = 2e-4 * np.sqrt(K) + 1e-5 * np.abs(ref).max()
    bad = np.abs(out - ref)
    assert bad.max() < tol, (bad.max(), tol, np.unravel_index(bad.argmax(), bad.shape))
    if prec == 2:       # split-bf16 x3: ~2^-16 per term, random signs -> a few 1e-5 * sqrt(K) on N(0,1) operands (float64 reference)
        ref64 = A.astype(np.float64) @ Bm.astype(np.float64).T + bias
        assert np.abs(out - ref64).max() < 6e-5 * np.sqrt(K), np.abs(out - ref64).max()


@pytest.mark.parametrize("shape", [(12500, 2048, 512), (9001, 3104, 160), (24577, 1024, 736), (3100, 4096, 128)])
def test_gemm_nt_mid_is_exact_on_small_integers(lib, shape):
    """The 128 x 256 two-workgroups-per-CU kernel (cn_gemm_nt_mid.hip; bf16, 128 <= K < 768 in whole k-tiles of 32, N <= 4096,
    >= 384 tiles of 256 x 256): operands and bias are small integers, so every product and every fp32 partial sum is exact and
    the result must EQUAL numpy's -- a row or k-chunk taken from the wrong place (the fill's XOR placement, the clamped edge
    rows, the last partial tile row / column) cannot hide inside a tolerance."""
    L, B = lib
    M, N, K = shape
    rng = np.random.RandomState(M + N + K)
    A = rng.randint(-3, 4, (M, K)).astype(np.float32); Bm = rng.randint(-3, 4, (N, K)).astype(np.float32)
    bias = rng.randint(-5, 6, N).astype(np.float32)
    out = np.zeros((M, N), np.float32)
    ctx = C.c_void_p(); B.check(L.cn_ctx_create(0, 1, None, C.byref(ctx)))
    try:
        B.check(L.cn_dbg_gemm_nt(ctx, A.ctypes.data, Bm.ctypes.data, out.ctypes.data, M, N, K, bias.ctypes.data, 2), ctx)
    finally:
        L.cn_ctx_destroy(ctx)
    ref = (A.astype(np.int64) @ Bm.astype(np.int64).T + bias.astype(np.int64)).astype(np.float32)
    assert np.array_equal(out, ref), np.argwhere(out != ref)[:4]


def _panel_everywhere(L, B, ctx):
    """By default the panel kernel takes K >= 512 on at most two column tiles and one round of CUs; the tests want it on every shape."""
    for name, value in ((b"nt_panel_min_ktiles", 1), (b"nt_panel_max_ntiles", 8), (b"nt_panel_max_panels", 384)):
        B.check(L.cn_ctx_set_option(ctx, name, value), ctx)


@pytest.mark.parametrize("flag", [0, 0x100, 0x200])
@pytest.mark.parametrize("with_bias", [True, False])
@pytest.mark.parametrize("shape", [(15000, 1024, 256), (14824, 256, 1024), (14824, 1024, 64), (15000, 192, 256), (15000, 256, 192),
                                   (61, 32, 64), (130, 1792, 128), (16001, 992, 320), (3000, 288, 1024)])
def test_gemm_nt_panel_is_exact_on_small_integers(lib, shape, with_bias, flag):
    """The one-panel-per-CU kernel (cn_gemm_nt_panel.hip; bf16, identity, K in whole k-tiles of 64, N <= 1792): shapes of the N-wide
    products of the headline step and edge cases (the last shape is what the default dispatch gives it).  Operands and bias are small integers, so every product and every fp32
    partial sum is exact and the result must EQUAL numpy's -- a chunk taken from the wrong place (the fill's XOR placement, the
    clamped edge rows and columns, a k-tile consumed before it landed: the counted vmcnt waits run across tile boundaries and
    count the stores of the tile before) cannot hide inside a tolerance.  flag: fp32 result / both results / the bf16 copy alone."""
    L, B = lib
    M, N, K = shape
    rng = np.random.RandomState(M + N + K)
    A = rng.randint(-3, 4, (M, K)).astype(np.float32); Bm = rng.randint(-3, 4, (N, K)).astype(np.float32)
    bias = rng.randint(-5, 6, N).astype(np.float32)
    if flag:            # (the copy is bf16: keep the sums exactly representable -- 8 bits)
        A = (A > 1).astype(np.float32); Bm = np.sign(Bm) * (np.abs(Bm) > 2); bias = np.sign(bias)
        Bm[:, 48:] = 0
    out = np.full((M, N), 7.0, np.float32)
    ctx = C.c_void_p(); B.check(L.cn_ctx_create(0, 1, None, C.byref(ctx)))
    try:
        _panel_everywhere(L, B, ctx)
        for rep in range(3):
            B.check(L.cn_dbg_gemm_nt(ctx, A.ctypes.data, Bm.ctypes.data, out.ctypes.data, M, N, K, bias.ctypes.data if with_bias else None, 2 | flag), ctx)
            ref = A.astype(np.int64) @ Bm.astype(np.int64).T + (bias.astype(np.int64) if with_bias else 0)
            assert np.abs(ref).max() < (256 if flag else 1 << 24)
            assert np.array_equal(out, ref.astype(np.float32)), (rep, np.argwhere(out != ref)[:4])
    finally:
        L.cn_ctx_destroy(ctx)


@pytest.mark.parametrize("shape", [(14824, 256, 1024), (15000, 512, 2048), (15000, 1024, 256)])
def test_gemm_nt_panel_is_repeatable(lib, shape):
    """Race screen of the panel kernel (fills in flight across barriers and tile boundaries on counted waits, per-wave staging
    reused tile after tile, LDS-DMA touches into a spare word): nothing in it is order dependent, so 40 launches on the same
    random operands must agree to the bit -- the first two shapes are what the default dispatch gives it (headline / reading B)."""
    L, B = lib
    M, N, K = shape
    rng = np.random.RandomState(M + N + K)
    A = rng.randn(M, K).astype(np.float32); Bm = rng.randn(N, K).astype(np.float32); bias = rng.randn(N).astype(np.float32)
    ctx = C.c_void_p(); B.check(L.cn_ctx_create(0, 1, None, C.byref(ctx)))
    try:
        _panel_everywhere(L, B, ctx)
        first = np.zeros((M, N), np.float32)
        B.check(L.cn_dbg_gemm_nt(ctx, A.ctypes.data, Bm.ctypes.data, first.ctypes.data, M, N, K, bias.ctypes.data, 2), ctx)
        ref = bf16_round(A) @ bf16_round(Bm).T + bias
        assert np.abs(first - ref).max() < 2e-4 * np.sqrt(K) + 1e-5 * np.abs(ref).max()
        out = np.zeros((M, N), np.float32)
        for rep in range(39):
            B.check(L.cn_dbg_gemm_nt(ctx, A.ctypes.data, Bm.ctypes.data, out.ctypes.data, M, N, K, bias.ctypes.data, 2), ctx)
            assert np.array_equal(out, first), (rep, np.abs(out - first).max())
    finally:
        L.cn_ctx_destroy(ctx)


def test_gemm_nt_panel_matches_the_tiled_kernel_bit_for_bit(lib):
    """Same k order, same MFMA (operands swapped): the panel kernel and gemm_nt_kernel agree to the bit on random operands."""
    L, B = lib
    M, N, K = 14824, 1024, 256
    rng = np.random.RandomState(5)
    A = rng.randn(M, K).astype(np.float32); Bm = rng.randn(N, K).astype(np.float32); bias = rng.randn(N).astype(np.float32)
    outs = []
    for off in (0, 1):
        out = np.zeros((M, N), np.float32)
        ctx = C.c_void_p(); B.check(L.cn_ctx_create(0, 1, None, C.byref(ctx)))
        try:
            _panel_everywhere(L, B, ctx)
            B.check(L.cn_ctx_set_option(ctx, b"no_nt_panel", off), ctx)
            B.check(L.cn_dbg_gemm_nt(ctx, A.ctypes.data, Bm.ctypes.data, out.ctypes.data, M, N, K, bias.ctypes.data, 2), ctx)
        finally:
            L.cn_ctx_destroy(ctx)
        outs.append(out)
    assert np.array_equal(outs[0], outs[1]), np.abs(outs[0] - outs[1]).max()


@pytest.mark.parametrize("with_bias", [True, False])
@pytest.mark.parametrize("shape", [(4200, 6176, 256), (9000, 3104, 320), (7000, 4128, 832)])
def test_gemm_nt_persistent_kernel_is_repeatable(lib, shape, with_bias, monkeypatch):
    """Race screen of the persistent 256 x 256 kernel (counted vmcnt waits, fills in flight across barriers and tile seams, two
    wave groups half a phase apart): nothing in it is order dependent, so 25 launches on the same operands must agree to the bit;
    K of 4, 5 and 13 k-tiles (the seam k-tiles are the first three of a tile; odd counts flip the buffer parity per tile).
    with_bias = False is the call the backward pass makes (K8 / K13: bias == NULL, identity, C only): the kernel still issues
    its two counted bias loads and must add exactly zero."""
    L, B = lib
    M, N, K = shape
    monkeypatch.setenv("CN_BIG8_MIN_K", "256")        # (by default products under a dozen k-tiles keep the plain 256 x 256 kernel)
    rng = np.random.RandomState(M + N + K)
    A = rng.randn(M, K).astype(np.float32); Bm = rng.randn(N, K).astype(np.float32); bias = rng.randn(N).astype(np.float32)
    bias_p = bias.ctypes.data if with_bias else None
    ctx = C.c_void_p(); B.check(L.cn_ctx_create(0, 1, None, C.byref(ctx)))
    try:
        first = np.zeros((M, N), np.float32)
        B.check(L.cn_dbg_gemm_nt(ctx, A.ctypes.data, Bm.ctypes.data, first.ctypes.data, M, N, K, bias_p, 2), ctx)
        ref = bf16_round(A) @ bf16_round(Bm).T + (bias if with_bias else 0.0)
        assert np.abs(first - ref).max() < 2e-4 * np.sqrt(K) + 1e-5 * np.abs(ref).max()
        out = np.zeros((M, N), np.float32)
        for rep in range(24 if with_bias else 4):
            B.check(L.cn_dbg_gemm_nt(ctx, A.ctypes.data, Bm.ctypes.data, out.ctypes.data, M, N, K, bias_p, 2), ctx)
            assert np.array_equal(out, first), (rep, np.abs(out - first).max())
    finally:
        L.cn_ctx_destroy(ctx)


@pytest.mark.parametrize("with_bias", [True, False])
@pytest.mark.parametrize("flag", [0x100, 0x200])
@pytest.mark.parametrize("act", [0, 2])
@pytest.mark.parametrize("shape", [(300, 256, 64), (6500, 4128, 512), (25000, 1024, 1024), (9001, 3104, 160)])
def test_gemm_nt_operand_copy_output(lib, shape, act, flag, with_bias, monkeypatch):
    """The operand-type (bf16) copy of the result, with and without the fp32 result beside it: the small shape runs the 128 x 128
    kernel, the large ones the persistent 256 x 256 kernel, whose seam stores count in its vmcnt waits (16 or 8 per phase)."""
    L, B = lib
    M, N, K = shape
    monkeypatch.setenv("CN_BIG8_MIN_K", "256")
    rng = np.random.RandomState(M + N + K + act)
    A = bf16_round(rng.randn(M, K)); Bm = bf16_round(rng.randn(N, K)); bias = rng.randn(N).astype(np.float32)
    out = np.zeros((M, N), np.float32)
    ctx = C.c_void_p(); B.check(L.cn_ctx_create(0, 1, None, C.byref(ctx)))
    try:
        B.check(L.cn_dbg_gemm_nt(ctx, A.ctypes.data, Bm.ctypes.data, out.ctypes.data, M, N, K,
                                 bias.ctypes.data if with_bias else None, act | flag), ctx)
    finally:
        L.cn_ctx_destroy(ctx)
    ref = A @ Bm.T + (bias if with_bias else 0.0)
    if act == 0:
        ref = np.tanh(ref)
    tol = 2e-4 * np.sqrt(K) + 2.0 ** -8 * np.abs(ref) + 1e-5          # (bf16 keeps 8 bits)
    bad = np.abs(out - ref) - tol
    assert bad.max() < 0, (bad.max(), np.unravel_index(bad.argmax(), bad.shape))


@pytest.mark.parametrize("shape", [(1000, 1024, 256), (6500, 4128, 512), (25000, 1024, 1024)])
def test_gemm_nt_without_bias(lib, shape):
    """bias == NULL on every nt kernel family (128 x 128, 256 x 256, persistent 256 x 256) at the default dispatch thresholds."""
    L, B = lib
    M, N, K = shape
    rng = np.random.RandomState(M + N + K + 1)
    A = rng.randn(M, K).astype(np.float32); Bm = rng.randn(N, K).astype(np.float32)
    out = np.full((M, N), 7.0, np.float32)
    ctx = C.c_void_p(); B.check(L.cn_ctx_create(0, 1, None, C.byref(ctx)))
    try:
        B.check(L.cn_dbg_gemm_nt(ctx, A.ctypes.data, Bm.ctypes.data, out.ctypes.data, M, N, K, None, 2), ctx)
    finally:
        L.cn_ctx_destroy(ctx)
    ref = bf16_round(A) @ bf16_round(Bm).T
    assert np.abs(out - ref).max() < 2e-4 * np.sqrt(K) + 1e-5 * np.abs(ref).max()


@pytest.mark.parametrize("prec", [0, 1, 2])
@pytest.mark.parametrize("shape", [(128, 32, 60), (1024, 256, 5000), (96, 160, 333), (4000, 2080, 700),
                                   (512, 64, 3000), (768, 96, 1111), (256, 416, 2000)])      # 256-row tiles: 256 x 64 (N < 128, partial N tile) and 256 x 128
def test_gemm_tn(lib, prec, shape):
    L, B = lib
    M, N, K = shape
    rng = np.random.RandomState(M + N + K)
    A = rng.randn(K, M).astype(np.float32); Bm = rng.randn(K, N).astype(np.float32)
    out = np.zeros((M, N), np.float32)
    ctx = C.c_void_p(); B.check(L.cn_ctx_create(0, prec, None, C.byref(ctx)))
    try:
        B.check(L.cn_dbg_gemm_tn(ctx, A.ctypes.data, Bm.ctypes.data, out.ctypes.data, M, N, K), ctx)
    finally:
        L.cn_ctx_destroy(ctx)
    if prec == 1:
        A, Bm = bf16_round(A), bf16_round(Bm)
    ref = A.T @ Bm
    tol = 2e-4 * np.sqrt(K) + 1e-5 * np.abs(ref).max()
    assert np.abs(out - ref).max() < tol
    if prec == 2:
        ref64 = A.astype(np.float64).T @ Bm.astype(np.float64)
        assert np.abs(out - ref64).max() < 6e-5 * np.sqrt(K), np.abs(out - ref64).max()


BIG_TN_SHAPES = [(512, 256, 4096),        # two whole tiles, whole k-tiles
                 (800, 448, 5003),        # M = 3 tiles + 32 rows, N = 256 + 192 (the narrowest last tile column that still goes big), K tail of 11 frames
                 (2048, 512, 9000),       # the LVCSR / reading-B layer product's shape (dW_in)
                 (1024, 256, 4111),       # ... and its dW_rec
                 (8000, 512, 6000)]       # LVCSR output layer: 31.25 tile rows


@pytest.mark.parametrize("shape", BIG_TN_SHAPES)
def test_gemm_tn_big(lib, shape, monkeypatch):
    """The 256 x 256 LDS-DMA gradient kernel (cn_gemm_tn_big.hip; bf16, M >= 512, N >= 192, K >= 4096): partial tile rows and
    columns, a K tail (frames past the end must come back from the fill as ZEROS -- buffer range check), several splits; against
    float64 numpy products of the rounded operands, and repeatable up to the order of its split-K atomics."""
    L, B = lib
    M, N, K = shape
    rng = np.random.RandomState(M + N + K)
    A = rng.randn(K, M).astype(np.float32); Bm = rng.randn(K, N).astype(np.float32)
    ctx = C.c_void_p(); B.check(L.cn_ctx_create(0, 1, None, C.byref(ctx)))
    try:
        outs = []
        for rep in range(3):
            out = np.zeros((M, N), np.float32)
            B.check(L.cn_dbg_gemm_tn(ctx, A.ctypes.data, Bm.ctypes.data, out.ctypes.data, M, N, K), ctx)
            outs.append(out)
    finally:
        L.cn_ctx_destroy(ctx)
    ref = bf16_round(A).astype(np.float64).T @ bf16_round(Bm).astype(np.float64)
    tol = 2e-5 * np.sqrt(K) + 1e-6 * np.abs(ref).max()          # fp32 accumulation of exact bf16 products
    for out in outs:
        bad = np.abs(out - ref)
        assert bad.max() < tol, (bad.max(), tol, np.unravel_index(bad.argmax(), bad.shape))
    # launches differ only in the order of the split-K atomics
    assert np.abs(outs[0] - outs[1]).max() < 1e-3 * tol + 2e-6 * np.abs(ref).max()


def test_gemm_tn_big_sees_every_frame_once(lib):
    """Unit impulses: A[k][m] = 1 only at m = k % M, B[k][n] = k-dependent small integers (exact in bf16): C[m][n] = the sum of
    B over the frames k = m (mod M) -- any frame read twice, skipped, or taken from a neighbouring split shows up exactly."""
    L, B = lib
    M, N, K = 512, 256, 4096 + 37
    A = np.zeros((K, M), np.float32); A[np.arange(K), np.arange(K) % M] = 1.0
    Bm = ((np.arange(K)[:, None] * 7 + np.arange(N)[None, :] * 3) % 13 - 6).astype(np.float32)
    out = np.zeros((M, N), np.float32)
    ctx = C.c_void_p(); B.check(L.cn_ctx_create(0, 1, None, C.byref(ctx)))
    try:
        B.check(L.cn_dbg_gemm_tn(ctx, A.ctypes.data, Bm.ctypes.data, out.ctypes.data, M, N, K), ctx)
    finally:
        L.cn_ctx_destroy(ctx)
    ref = A.T.astype(np.float64) @ Bm.astype(np.float64)
    assert np.array_equal(out, ref.astype(np.float32))
