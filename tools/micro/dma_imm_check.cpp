// check: does the immediate offset of global_load_lds_dwordx4 advance the LDS destination as well as the global
// source?  (LDS address = M0 + inst_offset + lane * 16 ?)  One wave, M0 written once, pieces at offsets
// -4096 .. 3072; the LDS image is copied out and compared with the source.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int OFF>
__device__ __forceinline__ void piece(const void *gsrc_uniform, uint32_t lane_off) {
  asm volatile("global_load_lds_dwordx4 %0, %1 offset:%2" : : "v"(lane_off), "s"(gsrc_uniform), "i"(OFF) : "memory");
}
__global__ void k(const unsigned char *src, unsigned char *out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x;
  for (int i = lane; i < 16384 / 4; i += 64) reinterpret_cast<uint32_t *>(smem)[i] = 0xdeadbeefu;
  __syncthreads();
  const uint32_t m0v = 4096 + 4096;     // LDS byte 8192 = the middle of the 8 pieces that start at LDS byte 4096
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" ::"s"(m0v) : "memory");
  const unsigned char *mid = src + 4096;
  piece<-4096>(mid, lane * 16u); piece<-3072>(mid, lane * 16u); piece<-2048>(mid, lane * 16u); piece<-1024>(mid, lane * 16u);
  piece<0>(mid, lane * 16u); piece<1024>(mid, lane * 16u); piece<2048>(mid, lane * 16u); piece<3072>(mid, lane * 16u);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = lane; i < 16384 / 4; i += 64) reinterpret_cast<uint32_t *>(out)[i] = reinterpret_cast<uint32_t *>(smem)[i];
}
int main() {
  std::vector<unsigned char> h(8192), o(16384);
  for (int i = 0; i < 8192; ++i) h[i] = (unsigned char)((i * 7 + (i >> 8)) & 0xff);
  unsigned char *d, *dout;
  hipMalloc(&d, 8192); hipMalloc(&dout, 16384);
  hipMemcpy(d, h.data(), 8192, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 16384, 0, d, dout);
  hipMemcpy(o.data(), dout, 16384, hipMemcpyDeviceToHost);
  int bad = 0, untouched = 0;
  for (int i = 0; i < 8192; ++i) bad += o[4096 + i] != h[i];
  for (int i = 0; i < 4096; ++i) untouched += (o[i] == ((0xdeadbeefu >> (8 * (i & 3))) & 0xff)) + (o[12288 + i] == ((0xdeadbeefu >> (8 * (i & 3))) & 0xff));
  printf("pieces at offsets -4096..3072 with one M0: %d wrong bytes of 8192; %d of 8192 guard bytes untouched -> %s\n", bad, untouched,
         bad == 0 && untouched == 8192 ? "the immediate moves BOTH addresses" : "NOT usable");
  return bad != 0;
}
