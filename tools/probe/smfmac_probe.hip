// Probe for v_smfmac_f32_16x16x64_bf16 on gfx950: (1) which A element (lane, slot, 2-bit index) meets which B element (lane, slot),
// (2) issue rate against the dense v_mfma_f32_16x16x32_bf16.   build: hipcc --offload-arch=gfx950 -O2 smfmac_probe.hip -o smfmac_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16v __attribute__((ext_vector_type(16)));
typedef float f4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// one wave per (la, ia, v): A has a single 1 at lane la slot ia whose index field is v; B slot codes: pass 0 -> lane+1, pass 1 -> slot+1
template <int ABID>
__global__ void map_kernel(float *out, int pass, int shift) {
    const int combo = blockIdx.x, la = combo >> 5, ia = (combo >> 2) & 7, v = combo & 3;
    const int lane = threadIdx.x;
    bf8 a; for (int i = 0; i < 8; ++i) a[i] = (__bf16)0.0f;
    if (lane == la) a[ia] = (__bf16)1.0f;
    unsigned idx = 0x4444u;                                   // (0,1) in every group
    idx = (idx & ~(3u << (2 * ia))) | ((unsigned)v << (2 * ia));
    idx <<= shift;
    bf16v b; for (int i = 0; i < 16; ++i) b[i] = (__bf16)(float)(pass == 0 ? lane + 1 : i + 1);
    f4 acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_smfmac_f32_16x16x64_bf16(a, b, acc, (int)idx, 0, ABID);
    for (int r = 0; r < 4; ++r) out[((size_t)combo * 64 + lane) * 4 + r] = acc[r];
}

template <int SPARSE, int NACC>
__global__ void __launch_bounds__(256) rate_kernel(float *out, long long *cycles, int iters) {
    bf8 a, b0; bf16v b;
    for (int i = 0; i < 8; ++i) b0[i] = (__bf16)(float)((threadIdx.x >> 3) & 3);
    for (int i = 0; i < 8; ++i) a[i] = (__bf16)(float)(threadIdx.x & 3);
    for (int i = 0; i < 16; ++i) b[i] = (__bf16)(float)((threadIdx.x >> 2) & 3);
    f4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (f4){0, 0, 0, 0};
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (SPARSE) acc[i % NACC] = __builtin_amdgcn_smfmac_f32_16x16x64_bf16(a, b, acc[i % NACC], 0x44444444, 0, 0);
            else acc[i % NACC] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b0, acc[i % NACC], 0, 0, 0);
        }
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0; for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

int main() {
    const int combos = 64 * 8 * 4;
    float *d; CK(hipMalloc(&d, (size_t)combos * 64 * 4 * sizeof(float)));
    std::vector<float> h0((size_t)combos * 256), h1(h0.size());
    for (int variant = 0; variant < 3; ++variant) {            // 0: abid 0, index in bits 15:0; 1: abid 1, bits 31:16; 2: abid 0, bits 31:16 (expect garbage)
        for (int pass = 0; pass < 2; ++pass) {
            if (variant == 0) map_kernel<0><<<combos, 64>>>(d, pass, 0);
            if (variant == 1) map_kernel<1><<<combos, 64>>>(d, pass, 16);
            if (variant == 2) map_kernel<0><<<combos, 64>>>(d, pass, 16);
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(pass ? h1.data() : h0.data(), d, h0.size() * sizeof(float), hipMemcpyDeviceToHost));
        }
        printf("== variant %d ==\n", variant);
        int shown = 0, bad = 0;
        for (int combo = 0; combo < combos; ++combo) {
            const int la = combo >> 5, ia = (combo >> 2) & 7, v = combo & 3;
            // D element (row, col): lane = col + 16 * (row / 4), reg = row % 4; look at col 0 and col 5
            int hits = 0, hrow = -1, lb = -1, ib = -1, lb5 = -1;
            for (int row = 0; row < 16; ++row) {
                float x = h0[((size_t)combo * 64 + 16 * (row / 4)) * 4 + row % 4];
                if (x != 0) { ++hits; hrow = row; lb = (int)x - 1; ib = (int)h1[((size_t)combo * 64 + 16 * (row / 4)) * 4 + row % 4] - 1;
                              lb5 = (int)h0[((size_t)combo * 64 + 5 + 16 * (row / 4)) * 4 + row % 4] - 1; }
            }
            // hypothesis: row = la & 15; with j = la >> 4: B lane group j' = 2 * (j & 1) + (ia >> 2), B slot = 8 * (j >> 1) + 4 * ((ia >> 1) & 1) + v
            const int j = la >> 4, jb = 2 * (j & 1) + (ia >> 2), sb = 8 * (j >> 1) + 4 * ((ia >> 1) & 1) + v;
            const bool ok = hits == 1 && hrow == (la & 15) && lb == 16 * jb && lb5 == 5 + 16 * jb && ib == sb;
            if (!ok) ++bad;
            if ((!ok && shown < 24) ) {
                printf("A lane %2d slot %d idx %d -> hits %d row %2d  B lane(col0) %2d (col5) %2d slot %2d %s\n", la, ia, v, hits, hrow, lb, lb5, ib, ok ? "" : "  <-- not the hypothesis");
                ++shown;
            }
        }
        printf("variant %d: %d of %d combos differ from the hypothesis (row = lane&15, B lane group 2*(j&1) + (slot>>2), B slot 8*(j>>1) + 4*((slot>>1)&1) + index)\n", variant, bad, combos);
    }

    // issue rate
    long long *dc; CK(hipMalloc(&dc, 1024 * sizeof(long long)));
    float *dout; CK(hipMalloc(&dout, 1024 * 256 * sizeof(float)));
    const int iters = 2000;
    for (int nacc : {8, 2, 1})
    for (int sparse = 0; sparse < 2; ++sparse)
        for (int blocks : {1, 256}) {
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            for (int rep = 0; rep < 2; ++rep) {
                CK(hipEventRecord(e0));
                if (nacc == 8) { if (sparse) rate_kernel<1, 8><<<blocks, 256>>>(dout, dc, iters); else rate_kernel<0, 8><<<blocks, 256>>>(dout, dc, iters); }
                if (nacc == 2) { if (sparse) rate_kernel<1, 2><<<blocks, 256>>>(dout, dc, iters); else rate_kernel<0, 2><<<blocks, 256>>>(dout, dc, iters); }
                if (nacc == 1) { if (sparse) rate_kernel<1, 1><<<blocks, 256>>>(dout, dc, iters); else rate_kernel<0, 1><<<blocks, 256>>>(dout, dc, iters); }
                CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
            }
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            long long c; CK(hipMemcpy(&c, dc, sizeof c, hipMemcpyDeviceToHost));
            printf("%s accumulators %d blocks %3d: %.3f ms, %.1f counter ticks per instruction (one wave per SIMD)\n", sparse ? "smfmac 16x16x64" : "mfma   16x16x32", nacc, blocks, ms,
                   c / (iters * 8.0));
        }
    return 0;
}
