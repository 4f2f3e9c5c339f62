// Does global_load_lds_dwordx4 accept an 8-byte aligned (not 16-byte aligned) source? (probe)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(64) void k(const float* in, float* out, int shift_floats, int stride_floats) {
  __shared__ f4 lds[64];
  const int lane = threadIdx.x;
  const float* src = in + shift_floats + (size_t)lane * stride_floats;
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds, 16, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  f4 v = lds[lane];
  out[4 * lane] = v.x; out[4 * lane + 1] = v.y; out[4 * lane + 2] = v.z; out[4 * lane + 3] = v.w;
}
int main() {
  const int N = 1 << 16;
  std::vector<float> h(N); for (int i = 0; i < N; ++i) h[i] = (float)i;
  float *in, *out; hipMalloc(&in, N * 4); hipMalloc(&out, 256 * 4);
  hipMemcpy(in, h.data(), N * 4, hipMemcpyHostToDevice);
  for (int shift : {0, 2, 1, 3}) for (int stride : {4, 16, 64}) {
    hipMemset(out, 0, 1024);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, in, out, shift, stride);
    hipError_t e = hipDeviceSynchronize();
    std::vector<float> o(256); hipMemcpy(o.data(), out, 1024, hipMemcpyDeviceToHost);
    int bad = 0; for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j) if (o[4 * l + j] != (float)(shift + l * stride + j)) ++bad;
    printf("shift %d floats (%2d B) stride %3d floats: %s, %d wrong words (lane1: %g %g %g %g)\n", shift, shift * 4, stride,
           hipGetErrorString(e), bad, o[4], o[5], o[6], o[7]);
  }
  return 0;
}
