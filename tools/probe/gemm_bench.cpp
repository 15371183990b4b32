// Times launch_gemm_nt / launch_gemm_tn on the shapes of the headline workload (bf16 operands, random data).
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -c tools/probe/gemm_bench.cpp -o /tmp/gb.o &&
//        hipcc --offload-arch=gfx950 /tmp/gb.o -Llstm-rnn_amd -lcurrennt_hip -Wl,-rpath,'$ORIGIN/../../lstm-rnn_amd' -o tools/probe/gemm_bench
// Options: the CN_<NAME> environment variables of cn_internal.h's option list (the library's process defaults: the probe has no context).
// GEMM_BENCH_NO_TN=1 skips the weight-gradient shapes.  GEMM_BENCH_COLD=1: every nt launch behind a cache flush, timed alone.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <string>
#include <cctype>

#include "../../lstm-rnn_amd/csrc/cn_internal.h"

using namespace cn;
#ifdef B8_STAMP
namespace cn { void b8_read_stamps(unsigned *h, unsigned *clk); }
static void print_stamps(int nk)
{
    unsigned h[2][2][4][5];
    unsigned clk[11];
    b8_read_stamps(&h[0][0][0][0], clk);
    printf("   k loops of workgroup 0: %u k-tiles, %u cycles in %.2f us = %.2f GHz, %.0f cycles per k-tile\n", clk[2], clk[0], clk[1] / 100.0, clk[0] / (clk[1] * 10.0), (double)clk[0] / clk[2]);
    printf("   cycles in normal / first / second / third k-tiles of a tile: group 0: %u %u %u %u, group 1: %u %u %u %u (sums over %d tiles)\n", clk[3], clk[4], clk[5], clk[6], clk[7], clk[8], clk[9], clk[10], (int)clk[2] / nk);
    nk = (int)clk[2];
    for (int w = 0; w < 2; ++w) for (int g = 0; g < 2; ++g) {
        printf("   wg %d group %d, s_memtime cycles per k-tile [wait, reads+barrier, operand wait, multiply, barrier]:", w ? 100 : 0, g);
        double tot = 0;
        for (int a = 0; a < 4; ++a) { printf("  ph%d:", a); for (int b = 0; b < 5; ++b) { double c = (double)h[w][g][a][b] / nk; tot += c; printf(" %5.0f", c); } }
        printf("  = %6.0f\n", tot);
    }
}
#endif

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

static void *rnd(size_t n16)   // n16 bf16 values, random small
{
    std::vector<unsigned short> h(n16);
    static const bool zero = getenv("GEMM_BENCH_ZERO") != nullptr;     // (zero operands draw less power: the clock stays up)
    for (size_t i = 0; i < n16; ++i) { float f = zero ? 0.f : (rand() % 2001 - 1000) / 1000.0f; unsigned u; memcpy(&u, &f, 4); h[i] = (unsigned short)(u >> 16); }
    void *d; CK(hipMalloc(&d, n16 * 2)); CK(hipMemcpy(d, h.data(), n16 * 2, hipMemcpyHostToDevice));
    return d;
}

int main()
{
    struct S { int M, N, K; const char *what; } nt[] = {
        {15600, 1024, 256, "input projection layer 2/3 (acts)"}, {15600, 1024, 64, "input projection layer 1"},
        {15600, 256, 1024, "error to preceding layer (a7)"}, {15600, 192, 256, "softmax projection"}, {15600, 256, 192, "softmax E_prev"},
        {15600, 2048, 512, "reading B input projection (Hp = 256)"}, {15600, 512, 2048, "reading B error to preceding layer"}, {32000, 4096, 1024, "long-utterance input projection (Hp = 512)"},
        {25600, 8000, 1024, "LVCSR softmax projection"}, {25600, 1024, 8000, "LVCSR softmax E_prev"}, {25600, 2048, 1024, "LVCSR layer input projection (Hp = 256)"},
        {51200, 2048, 512, "LVCSR as run by bench.py: input projection of layers 2-4 (blstm512 = 256 per direction), T = 800"},
        {35200, 2048, 512, "the same at T = 550"}, {51200, 512, 2048, "LVCSR error to the preceding layer"}, {51200, 8000, 512, "LVCSR softmax projection"}, {51200, 512, 8000, "LVCSR softmax E_prev"}};
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    // GEMM_BENCH_MDIV=<d>: every M (nt) / K (tn) divided by d -- how the products scale with the number of frames
    const int mdiv = getenv("GEMM_BENCH_MDIV") ? atoi(getenv("GEMM_BENCH_MDIV")) : 1;
    const int nshapes = getenv("GEMM_BENCH_FIRST") ? atoi(getenv("GEMM_BENCH_FIRST")) : 1000;
    int ishape = 0;
    for (auto &c : nt) {
        if (ishape++ >= nshapes) break;
        c.M /= mdiv;
        void *A = rnd((size_t)c.M * c.K), *B = rnd((size_t)c.N * c.K);
        float *C, *bias; CK(hipMalloc((void **)&C, (size_t)c.M * c.N * 4)); CK(hipMalloc((void **)&bias, c.N * 4)); CK(hipMemset(bias, 0, c.N * 4));
        GemmNT g{}; g.A = A; g.lda = c.K; g.B = B; g.ldb = c.K; g.C = C; g.ldc = c.N; g.bias = bias; g.act = ACT_IDENTITY; g.M = c.M; g.N = c.N; g.K = c.K;
        for (int i = 0; i < 3; ++i) launch_gemm_nt(s, false, g);
        const int reps = 20;
        float ms;
        if (getenv("GEMM_BENCH_COLD")) {
            // every launch behind a 600 MB fill (L2s and Infinity Cache hold nothing of the operands: the situation inside a
            // training step, where 1.4 GB pass between two uses of anything), timed alone
            static char *flush = nullptr;
            if (!flush) CK(hipMalloc((void **)&flush, 600u << 20));
            ms = 0.f;
            for (int i = 0; i < reps; ++i) {
                CK(hipMemsetAsync(flush, i, 600u << 20, s));
                CK(hipEventRecord(e0, s));
                launch_gemm_nt(s, false, g);
                CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
                float one; CK(hipEventElapsedTime(&one, e0, e1)); ms += one;
            }
        } else {
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < reps; ++i) launch_gemm_nt(s, false, g);
        CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        }
        double us = ms * 1e3 / reps, bytes = (double)c.M * c.N * 4 + (double)c.M * c.K * 2 + (double)c.N * c.K * 2, fl = 2.0 * c.M * c.N * c.K;
        printf("gemm_nt M=%5d N=%4d K=%4d  %7.1f us  %6.0f GB/s (compulsory bytes)  %6.1f TFLOP/s   %s\n", c.M, c.N, c.K, us, bytes / us * 1e-3, fl / us * 1e-6, c.what);
#ifdef B8_STAMP
        if (gemm_nt_big_applies(false, g)) print_stamps(c.K / 64);
#endif
        CK(hipFree(A)); CK(hipFree(B)); CK(hipFree(C)); CK(hipFree(bias));
    }
    struct S tn[] = {{1024, 256, 15600, "dW_in layer 2/3"}, {1024, 64, 15600, "dW_in layer 1"}, {512, 128, 15548, "dW_rec per direction"}, {192, 256, 15600, "softmax dW"}, {8000, 1024, 25600, "LVCSR softmax dW"},
                   {2048, 512, 35200, "LVCSR dW_in of layers 2-4 (blstm512), T = 550"}, {1024, 256, 35136, "LVCSR dW_rec per direction"}, {8000, 512, 35200, "LVCSR softmax dW as run by bench.py"},
                   {2048, 512, 15600, "reading B dW_in (Hp = 256)"}, {1024, 256, 15548, "reading B dW_rec per direction"}, {4096, 1024, 32000, "long-utterance dW_in (Hp = 512)"}};
    if (getenv("GEMM_BENCH_NO_TN")) return 0;
    const int tn_first = getenv("GEMM_BENCH_TN_FIRST") ? atoi(getenv("GEMM_BENCH_TN_FIRST")) : 0;      // skip the first n tn shapes
    int itn = 0;
    for (auto &c : tn) {
        if (itn++ < tn_first) continue;
        c.K /= mdiv;
        void *A = rnd((size_t)c.K * c.M), *B = rnd((size_t)c.K * c.N);
        float *C; CK(hipMalloc((void **)&C, (size_t)c.M * c.N * 4)); CK(hipMemset(C, 0, (size_t)c.M * c.N * 4));
        GemmTN g{}; g.A = A; g.lda = c.M; g.B = B; g.ldb = c.N; g.C = C; g.ldc = c.N; g.M = c.M; g.N = c.N; g.K = c.K;
        for (int i = 0; i < 3; ++i) launch_gemm_tn(s, false, g);
        CK(hipEventRecord(e0, s));
        const int reps = 20;
        for (int i = 0; i < reps; ++i) launch_gemm_tn(s, false, g);
        CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        double us = ms * 1e3 / reps, bytes = (double)c.K * (c.M + c.N) * 2, fl = 2.0 * c.M * c.N * c.K;
        printf("gemm_tn M=%5d N=%4d K=%5d  %7.1f us  %6.0f GB/s (operand bytes)  %6.1f TFLOP/s   %s\n", c.M, c.N, c.K, us, bytes / us * 1e-3, fl / us * 1e-6, c.what);
        CK(hipFree(A)); CK(hipFree(B)); CK(hipFree(C));
    }
    // the grouped launch of an LSTM layer's backward pass: dW_in + dW_rec per direction in ONE launch (cn_api.cpp: lstm_backward)
    struct Grp { int P, K, Hp, PS; const char *what; } grps[] = {{256, 15600, 128, 52, "layer 2/3 group (dW_in 1024x256 + 2 x dW_rec 512x128)"}, {64, 15600, 128, 52, "layer 1 group (dW_in 1024x64 + 2 x dW_rec 512x128)"},
        {512, 35200, 256, 64, "LVCSR layer 2-4 group (dW_in 2048x512 + 2 x dW_rec 1024x256), T = 550"}, {64, 35200, 256, 64, "LVCSR layer 1 group (dW_in 2048x64 + 2 x dW_rec 1024x256)"},
        {512, 15600, 256, 52, "reading B layer 2/3 group"}, {64, 15600, 256, 52, "reading B layer 1 group (dW_in 2048x64 + 2 x dW_rec 1024x256): the exposed tail"},
        {64, 51200, 256, 64, "LVCSR layer 1 group at T = 800: the exposed tail"}, {1024, 32000, 512, 16, "long-utterance layer 2-5 group (dW_in 4096x1024 + 2 x dW_rec 2048x512)"}};
    for (auto &gr : grps) {
        const int K = gr.K / mdiv, Hp = gr.Hp, R = 8 * Hp, PS = gr.PS;
        void *delta = rnd((size_t)K * R), *x = rnd((size_t)K * gr.P), *y = rnd((size_t)K * 2 * Hp);
        float *C; const size_t cfl = (size_t)R * gr.P + (size_t)R * Hp; CK(hipMalloc((void **)&C, cfl * 4)); CK(hipMemset(C, 0, cfl * 4));
        GemmTN gs[3] = {};
        gs[0].A = delta; gs[0].lda = R; gs[0].B = x; gs[0].ldb = gr.P; gs[0].C = C; gs[0].ldc = gr.P; gs[0].M = R; gs[0].N = gr.P; gs[0].K = K;
        for (int d = 0; d < 2; ++d) {
            GemmTN &r = gs[1 + d];
            r.A = (char *)delta + (size_t)d * 4 * Hp * 2 + (d == 0 ? (size_t)PS * R * 2 : 0);
            r.B = (char *)y + (size_t)d * Hp * 2 + (d == 1 ? (size_t)PS * 2 * Hp * 2 : 0);
            r.lda = R; r.ldb = 2 * Hp; r.C = C + (size_t)R * gr.P + (size_t)d * 4 * Hp * Hp; r.ldc = Hp; r.M = 4 * Hp; r.N = Hp; r.K = K - PS;
        }
        for (int i = 0; i < 3; ++i) launch_gemm_tn_group(s, false, gs, 3);
        CK(hipEventRecord(e0, s));
        const int reps = 20;
        for (int i = 0; i < reps; ++i) launch_gemm_tn_group(s, false, gs, 3);
        CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        double us = ms * 1e3 / reps, bytes = (double)K * (R + gr.P + 2 * Hp) * 2, fl = 2.0 * K * ((double)R * gr.P + (double)R * Hp);
        printf("gemm_tn group  %7.1f us  %6.0f GB/s (unique operand bytes)  %6.1f TFLOP/s   %s\n", us, bytes / us * 1e-3, fl / us * 1e-6, gr.what);
        CK(hipFree(delta)); CK(hipFree(x)); CK(hipFree(y)); CK(hipFree(C));
    }
    return 0;
}
