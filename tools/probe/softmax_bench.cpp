// Times the wide-row softmax launchers on the LVCSR output layer's shape (35 200 patterns x 8000 classes, ragged: ~30 % of the
// patterns are padding) beside two plain streaming kernels over the same buffer (what the memory system gives a read pass and a
// read + half-width write pass).
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -c tools/probe/softmax_bench.cpp -o /tmp/sb.o &&
//        hipcc --offload-arch=gfx950 /tmp/sb.o lstm-rnn_amd/csrc/cn_elementwise.o -o tools/probe/softmax_bench
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../lstm-rnn_amd/csrc/cn_internal.h"

using namespace cn;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef __attribute__((ext_vector_type(4))) float f32x4;
__global__ __launch_bounds__(256) void stream_read_kernel(const float *y, const char *pat, int N, int Lp, float *out)
{
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (long row = blockIdx.x; row < N; row += gridDim.x) {
        if (pat[row] == 0) continue;
        const float *r = y + row * Lp;
        for (int j = 4 * threadIdx.x; j < Lp; j += 1024) acc += *(const f32x4 *)(r + j);
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) out[0] = 1.f;
}
__global__ __launch_bounds__(256) void stream_rw_kernel(const float *y, const char *pat, int N, int Lp, unsigned short *out)
{
    typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
    for (long row = blockIdx.x; row < N; row += gridDim.x) {
        const bool real = pat[row] != 0;
        const float *r = y + row * Lp;
        for (int j = 4 * threadIdx.x; j < Lp; j += 1024) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (real) v = *(const f32x4 *)(r + j);
            *(bf16x4 *)((__bf16 *)out + row * Lp + j) = bf16x4{(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
        }
    }
}

// argv[1] = "narrow": the headline's output layer instead (183 classes, 50 sequences of 250-350 frames)
int main(int argc, char **argv)
{
    const bool narrow = argc > 1 && !strcmp(argv[1], "narrow");
    const int PS = narrow ? 50 : 64, PSp = narrow ? 52 : 64, L = narrow ? 183 : 8000, Lp = narrow ? 192 : 8000;
    srand(7);
    std::vector<int> len(PSp, 0);
    int T = 0;
    for (int s = 0; s < PS; ++s) { len[s] = narrow ? 250 + rand() % 101 : 300 + rand() % 501; if (len[s] > T) T = len[s]; }
    const int N = T * PSp;
    std::vector<char> pat(N); std::vector<int> tc(N);
    long real = 0;
    for (int t = 0; t < T; ++t) for (int s = 0; s < PSp; ++s) { const bool r = t < len[s]; pat[t * PSp + s] = r ? 1 : 0; tc[t * PSp + s] = r ? rand() % L : -1; real += r; }
    printf("N = %d patterns (%ld real = %.1f %%), L = %d: logits %.2f GB, real rows %.2f GB\n", N, real, 100.0 * real / N, L, N * (double)Lp * 4 / 1e9, real * (double)Lp * 4 / 1e9);
    float *y, *y0, *rowstat, *smstat, *colsum, *out; char *dpat; int *dtc; void *delta;
    CK(hipMalloc(&y, (size_t)N * Lp * 4)); CK(hipMalloc(&y0, (size_t)N * Lp * 4));
    CK(hipMalloc(&rowstat, N * 8)); CK(hipMalloc(&smstat, N * 8)); CK(hipMalloc(&colsum, Lp * 4)); CK(hipMalloc(&out, 64));
    CK(hipMalloc(&dpat, N)); CK(hipMalloc(&dtc, N * 4)); CK(hipMalloc(&delta, (size_t)N * Lp * 2));
    {
        std::vector<float> h((size_t)N * Lp);
        for (size_t i = 0; i < h.size(); ++i) h[i] = (rand() % 20001 - 10000) / 2000.0f;       // logits in [-5, 5]
        CK(hipMemcpy(y0, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    }
    CK(hipMemcpy(dpat, pat.data(), N, hipMemcpyHostToDevice)); CK(hipMemcpy(dtc, tc.data(), N * 4, hipMemcpyHostToDevice));
    CK(hipMemset(colsum, 0, Lp * 4));
    float *loss2; CK(hipMalloc(&loss2, 64 * 4)); CK(hipMemset(loss2, 0, 64 * 4));
    float *colpart; CK(hipMalloc(&colpart, softmax_mcc_bwd_colpart_floats() * 4)); CK(hipMemset(colpart, 0, softmax_mcc_bwd_colpart_floats() * 4));
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const double gb_row = real * (double)Lp * 4 / 1e9, gb_half = N * (double)Lp * 2 / 1e9;
    auto timeit = [&](const char *name, double gbytes, auto &&f, bool restore) {
        float best = 1e9f, sum = 0.f; const int reps = narrow ? 21 : 6;
        for (int i = 0; i < reps; ++i) {
            if (restore) CK(hipMemcpyAsync(y, y0, (size_t)N * Lp * 4, hipMemcpyDeviceToDevice, s));
            CK(hipEventRecord(a, s)); f(); CK(hipEventRecord(b, s)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            if (i > 0) { sum += ms; if (ms < best) best = ms; }
        }
        printf("%-44s %8.1f us (best %8.1f)  %6.3f GB -> %5.2f TB/s\n", name, 1e3 * sum / (reps - 1), 1e3 * best, gbytes, gbytes / (sum / (reps - 1)));
    };
    CK(hipMemcpy(y, y0, (size_t)N * Lp * 4, hipMemcpyDeviceToDevice));
    if (narrow) {       // (a launch with nothing to do: what an event-timed launch costs by itself)
        timeit("empty rows (N = 0 patterns real)", 0.0, [&] { hipLaunchKernelGGL(stream_read_kernel, dim3(256), dim3(256), 0, s, y, dpat, 0, Lp, out); }, false);
    }
    for (int blocks : {1024, 2048, 4096})
        timeit(("stream read, " + std::to_string(blocks) + " workgroups").c_str(), gb_row, [&] { hipLaunchKernelGGL(stream_read_kernel, dim3(blocks), dim3(256), 0, s, y, dpat, N, Lp, out); }, false);
    for (int blocks : {1024, 2048, 4096})
        timeit(("stream read + bf16 write, " + std::to_string(blocks) + " workgroups").c_str(), gb_row + gb_half, [&] { hipLaunchKernelGGL(stream_rw_kernel, dim3(blocks), dim3(256), 0, s, y, dpat, N, Lp, (unsigned short *)delta); }, false);
    for (int fast = 0; fast < 2; ++fast) {
        timeit(fast ? "softmax fwd eager fast" : "softmax fwd eager exact", 2 * gb_row, [&] { launch_softmax_fwd(s, y, dpat, N, L, Lp, dtc, rowstat, fast, nullptr); }, true);
        timeit(fast ? "softmax bwd eager, column sums in replicas" : "softmax bwd eager (on posteriors)", gb_row + gb_half, [&] { launch_softmax_mcc_bwd(s, false, y, dtc, dpat, N, L, Lp, nullptr, delta, colsum, nullptr, nullptr, nullptr, nullptr, false, fast ? colpart : nullptr); }, false);
        if (narrow) {
            if (fast) timeit("softmax bwd eager, replicas + loss sum (16 workgroups)", gb_row + gb_half, [&] { launch_softmax_mcc_bwd(s, false, y, dtc, dpat, N, L, Lp, nullptr, delta, colsum, rowstat, loss2, loss2 + 2, nullptr, false, colpart); }, false);
            continue;
        }
        timeit(fast ? "softmax fwd lazy fast" : "softmax fwd lazy exact", gb_row, [&] { launch_softmax_fwd(s, y, dpat, N, L, Lp, dtc, rowstat, fast, smstat); }, true);
        timeit(fast ? "softmax bwd lazy fast" : "softmax bwd lazy exact", gb_row + gb_half, [&] { launch_softmax_mcc_bwd(s, false, y, dtc, dpat, N, L, Lp, nullptr, delta, colsum, nullptr, nullptr, nullptr, smstat, fast); }, false);
    }
    return 0;
}
