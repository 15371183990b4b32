#include <hip/hip_runtime.h>
typedef __attribute__((ext_vector_type(4))) float f32x4;
__device__ __forceinline__ void dma16(unsigned lds_dst, const void *sbase, unsigned voff)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void dma4(unsigned lds_dst, const void *sbase, unsigned voff)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void dma1(unsigned lds_dst, const void *sbase, unsigned voff)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_ubyte %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}
__global__ void k(const float* src, float* dst, int n) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  const float* s0 = src + blockIdx.x * 1024;
  dma16(base + wave * 1024, s0, (wave * 64 + lane) * 16);
  dma4(base + 8192 + wave * 256, s0, (wave * 64 + lane) * 4);
  dma1(base + 12288 + wave * 256, s0, (wave * 64 + lane));
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  f32x4 v = *(const f32x4*)(smem + wave * 1024 + lane * 16);
  float w = *(const float*)(smem + 8192 + wave * 256 + lane * 4);
  unsigned b = *(const unsigned*)(smem + 12288 + wave * 256 + lane * 4);
  dst[blockIdx.x * 256 + threadIdx.x] = v[0] + v[1] + v[2] + v[3] + w + b;
}
#include <cstdio>
#include <vector>
int main() {
  const int nb = 4;
  std::vector<float> h(nb * 1024);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)(i % 97) * 0.5f;
  float *src, *dst;
  hipMalloc(&src, h.size() * 4); hipMalloc(&dst, nb * 256 * 4);
  hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(nb), dim3(256), 16384, 0, src, dst, 0);
  std::vector<float> o(nb * 256);
  hipMemcpy(o.data(), dst, o.size() * 4, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int b = 0; b < nb; ++b) for (int t = 0; t < 256; ++t) {
    const float *s0 = &h[b * 1024];
    const unsigned char *bytes = (const unsigned char *)s0;
    float ref = s0[t * 4] + s0[t * 4 + 1] + s0[t * 4 + 2] + s0[t * 4 + 3] + s0[t] + bytes[t];
    if (ref != o[b * 256 + t]) { if (bad < 5) printf("mismatch b=%d t=%d ref=%f got=%f\n", b, t, ref, o[b * 256 + t]); ++bad; }
  }
  printf("lds dma probe: %d mismatches\n", bad);
  return bad != 0;
}
