// fma_mix_probe.hip -- v_fma_mix_f32 (fp16 operand converted inside the FMA) against v_cvt_f32_f16 + fma: same bits, subnormals included?
//   hipcc --offload-arch=gfx950 -O3 tools/probes/fma_mix_probe.hip -o /tmp/fma_mix_probe && /tmp/fma_mix_probe
#include <hip/hip_runtime.h>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
// sum[q] += w * (float)v[q] with v_fma_mix_f32: the fp16 operand is converted inside the instruction (one op per element instead of cvt + fma)
__device__ __forceinline__ void fma_mix8(float (&sum)[8], float w, half8 v) {
  const floatx4 p = __builtin_bit_cast(floatx4, v);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[0,1,0]" : "+v"(sum[2 * k]) : "v"(w), "v"(p[k]));
    asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[0,1,0]" : "+v"(sum[2 * k + 1]) : "v"(w), "v"(p[k]));
  }
}
__global__ void k(const half8* x, const float* w, float* y, float* y2) {
  float s[8] = {0,0,0,0,0,0,0,0}, s2[8] = {0,0,0,0,0,0,0,0};
  for (int i = 0; i < 4; ++i) {
    half8 v = x[threadIdx.x * 4 + i]; float ww = w[i];
    fma_mix8(s, ww, v);
    for (int q = 0; q < 8; ++q) s2[q] += ww * (float)v[q];
  }
  for (int q = 0; q < 8; ++q) { y[threadIdx.x * 8 + q] = s[q]; y2[threadIdx.x * 8 + q] = s2[q]; }
}
int main() {
  half8 *x; float *w, *y, *y2;
  hipMalloc(&x, 64*4*16); hipMalloc(&w, 16); hipMalloc(&y, 64*8*4); hipMalloc(&y2, 64*8*4);
  _Float16 hx[64*4*8]; float hw[4] = {0.37f, 1.25f, 0.001f, 3.f};
  unsigned s = 1; for (int i = 0; i < 64*4*8; ++i) { s = s * 1664525u + 1013904223u; hx[i] = (_Float16)(((int)(s >> 16) % 2000 - 1000) / 317.f); if (i % 7 == 3) hx[i] = (_Float16)(((int)(s >> 20) % 60 - 30) * 6e-8f); }   // (every seventh: fp16 subnormals)
  hipMemcpy(x, hx, sizeof hx, hipMemcpyHostToDevice); hipMemcpy(w, hw, 16, hipMemcpyHostToDevice);
  k<<<1, 64>>>(x, w, y, y2);
  float a[512], b[512]; hipMemcpy(a, y, 2048, hipMemcpyDeviceToHost); hipMemcpy(b, y2, 2048, hipMemcpyDeviceToHost);
  int bad = 0; for (int i = 0; i < 512; ++i) bad += a[i] != b[i];
  printf("fma_mix vs cvt+fma: %d of 512 differ (e.g. %g %g)\n", bad, a[5], b[5]);
  return bad != 0;
}
