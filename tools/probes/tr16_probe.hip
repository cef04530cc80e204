// Probe of ds_read_b64_tr_b16 on gfx950: which LDS elements does lane l receive when lane l passes the address
// of elements [4l, 4l+4)?  Prints lane -> 4 element indices.  Build+run on the GPU box:
//   hipcc --offload-arch=gfx950 -O2 tools/probes/tr16_probe.hip -o /tmp/tr16_probe && /tmp/tr16_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short v4s __attribute__((vector_size(8)));
__global__ void k(short *out, int stride_elems) {
  __shared__ __attribute__((aligned(16))) short lds[8192];
  for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = (short)i;
  __syncthreads();
  // lane (g = l>>4, i = l&15) points at row g*4 + i/4, col (i%4)*4 of a matrix with `stride_elems` per row
  const int l = threadIdx.x, g = l >> 4, i = l & 15;
  short *p = lds + (g * 4 + (i >> 2)) * stride_elems + (i & 3) * 4;
  v4s v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s *)p);
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = v[j];
}
int main() {
  short *d, h[256];
  (void)hipMalloc(&d, sizeof(h));
  for (int stride : {16, 128}) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, stride);
    (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("row stride %d elements: lane -> (row,col) x4\n", stride);
    for (int l = 0; l < 64; ++l) {
      printf("lane %2d:", l);
      for (int j = 0; j < 4; ++j) printf(" (%d,%d)", h[l * 4 + j] / stride, h[l * 4 + j] % stride);
      printf("\n");
    }
  }
  return 0;
}
