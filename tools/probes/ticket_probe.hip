// ticket_probe.hip -- what does a "last-arriving workgroup finalizes" tail cost on MI355X (8 XCDs, one L2 each)?
//
// The BatchNorm statistics of a convolution's output leave its epilogue as per-row-tile partial sums; a separate ~5.5 us kernel
// (bn_finalize_kernel, 90 + 90 launches per training step) turns them into scale / shift.  Folding that kernel into the tail of
// the LAST workgroup of each column of tiles needs, in every workgroup: partial stores -> agent-scope release fence -> one integer
// atomic on a ticket; and in the last one: acquire + the reduction.  On gfx950 the L2s of different XCDs are not coherent with
// each other for ordinary memory, so the release is a `buffer_wbl2 sc1` (write back this XCD's dirty lines) -- which in a
// convolution epilogue means "wait until the tile this workgroup just stored has left L2".  This probe measures exactly that:
//
//   variant 0  producer kernel (each workgroup stores a 160 x 128 fp16 tile = 40 KB + its partials)  +  separate finalize kernel
//   variant 1  producer kernel with fence + ticket + last-block reduction (no second launch)
//   variant 2  variant 1 without the reduction (fence + ticket only: the fence's own cost)
//   variant 3  (round 6) NO fence: the partials themselves leave as agent-scope relaxed atomic stores (global_store ... sc1: written
//              through this XCD's L2), s_waitcnt vmcnt(0), workgroup barrier, relaxed agent-scope ticket; the last workgroup reads the
//              column's partials with agent-scope atomic loads (sc1: not served from its own L2).  The 40 KB tile is NOT waited for.
//   variant 4  variant 3 without the reduction
//
// printed: microseconds per (producer [+ finalize]) pair, averaged over many back-to-back launches on one stream, for the grid
// shapes of the step's stage-3 layers (256 / 1024 tiles) and a check that variant 1's sums equal variant 0's.
//
//   hipcc --offload-arch=gfx950 -O3 tools/probes/ticket_probe.hip -o /tmp/ticket_probe && /tmp/ticket_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef _Float16 half_t;
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

constexpr int BM = 160, BN = 128, T = 256;

// VAR 0: store tile + partials.  VAR 1: + fence, ticket, last block of the column reduces.  VAR 2: + fence, ticket only.
template <int VAR>
__global__ __launch_bounds__(T) void producer(half_t *y, float *part, int mtiles, int ntiles, int C, unsigned *ticket, float *fin, int work, float seed = 0.f) {
  const int lin = blockIdx.x, xcd = lin & 7, j = lin >> 3;
  const int nt = j % ntiles, mt = (j / ntiles) * 8 + xcd;
  if (mt >= mtiles) return;
  const int tid = threadIdx.x;
  // some arithmetic so that the tile is not a pure store burst (the K loop's stand-in)
  float a = (float)(tid + mt), b = 1.0001f;
  for (int k = 0; k < work; ++k) a = a * b + 0.5f;
  // the output tile: 160 rows x 128 channels fp16, 16 bytes per lane, rows of the full tensor are C channels wide
  half8 v;
  for (int r = 0; r < 8; ++r) v[r] = (half_t)(a * 1e-6f + r);
  for (int idx = tid; idx < BM * (BN / 8); idx += T) {
    const int row = idx / (BN / 8), ch = idx - row * (BN / 8);
    *reinterpret_cast<half8 *>(y + ((size_t)(mt * BM + row) * C + nt * BN + ch * 8)) = v;
  }
  // partials [mt][2][C]
  if (tid < BN) {
    part[((size_t)mt * 2 + 0) * C + nt * BN + tid] = (float)(mt + 1) * 0.25f + tid + seed;
    part[((size_t)mt * 2 + 1) * C + nt * BN + tid] = (float)(mt + 1) * 0.5f;
  }
  if (VAR == 0) return;
  __shared__ unsigned last;
  if (VAR >= 3) {
    // (the plain stores above are overwritten here with the same values through the coherent path; a real kernel would only issue these)
    if (tid < BN) {
      __hip_atomic_store(&part[((size_t)mt * 2 + 0) * C + nt * BN + tid], (float)(mt + 1) * 0.25f + tid + seed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&part[((size_t)mt * 2 + 1) * C + nt * BN + tid], (float)(mt + 1) * 0.5f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      const unsigned t = __hip_atomic_fetch_add(&ticket[nt], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      last = (t == (unsigned)mtiles - 1u) ? 1u : 0u;
      if (last) __hip_atomic_store(&ticket[nt], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (!last || VAR == 4) return;
    const int c = tid & (BN - 1), which = tid >> 7;
    double s = 0.0;
    for (int k0 = 0; k0 < mtiles; k0 += 16) {      // sixteen loads in flight (hand-issued: hipcc waits after every atomic load), summed in tile order
      float v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const float *q = &part[((size_t)(k0 + u < mtiles ? k0 + u : 0) * 2 + which) * C + nt * BN + c];
        asm volatile("global_load_dword %0, %1, off sc1" : "=v"(v[u]) : "v"(q) : "memory");
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        asm volatile("" : "+v"(v[u]));
        if (k0 + u < mtiles) s += (double)v[u];
      }
    }
    fin[(size_t)which * C + nt * BN + c] = (float)s;
    return;
  }
  __threadfence();                      // agent-scope release: this workgroup's stores are visible to the other XCDs
  __syncthreads();
  if (tid == 0) {
    const unsigned t = atomicAdd(&ticket[nt], 1u);
    last = (t == (unsigned)mtiles - 1u) ? 1u : 0u;
    if (last) ticket[nt] = 0;           // self-cleaning: the next launch starts from zero again (graph replay needs no memset)
  }
  __syncthreads();
  if (!last || VAR == 2) return;
  __threadfence();                      // acquire side
  // the last workgroup of column nt: sum the column's partials in tile order (fixed order -> deterministic), in double
  const int c = tid & (BN - 1), which = tid >> 7;   // 256 threads: 128 channels x {sum, sumsq}
  double s = 0.0;
  for (int k = 0; k < mtiles; ++k) s += (double)__builtin_nontemporal_load(&part[((size_t)k * 2 + which) * C + nt * BN + c]);
  fin[(size_t)which * C + nt * BN + c] = (float)s;
}

__global__ __launch_bounds__(1024) void finalize(const float *part, int nblk, int C, float *fin) {
  __shared__ double red[2][32][33];
  const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  double s0 = 0.0, s1 = 0.0;
  if (c < C)
    for (int k = rl; k < nblk; k += 32) {
      s0 += (double)part[(size_t)k * 2 * C + c];
      s1 += (double)part[(size_t)k * 2 * C + C + c];
    }
  red[0][rl][cl] = s0;
  red[1][rl][cl] = s1;
  __syncthreads();
  if (rl == 0 && c < C) {
    double a = 0, b = 0;
    for (int k = 0; k < 32; ++k) { a += red[0][k][cl]; b += red[1][k][cl]; }
    fin[c] = (float)a;
    fin[C + c] = (float)b;
  }
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

int main() {
  const int M = 20480;
  const int mtiles = M / BM;
  for (int C : {256, 1024}) {
    for (int work : {2000, 20000}) {
      const int ntiles = C / BN;
      half_t *y; float *part, *fin0, *fin1; unsigned *ticket;
      CK(hipMalloc(&y, (size_t)M * C * 2));
      CK(hipMalloc(&part, (size_t)mtiles * 2 * C * 4));
      CK(hipMalloc(&fin0, 2 * C * 4)); CK(hipMalloc(&fin1, 2 * C * 4));
      CK(hipMalloc(&ticket, 64 * 4)); CK(hipMemset(ticket, 0, 64 * 4));
      const dim3 grid((mtiles + 7) / 8 * 8 * ntiles);
      hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      const int reps = 200;
      float us[5];
      for (int var = 0; var < 5; ++var) {
        for (int it = 0; it < reps + 20; ++it) {
          if (it == 20) CK(hipEventRecord(e0, 0));
          if (var == 0) {
            hipLaunchKernelGGL(producer<0>, grid, dim3(T), 0, 0, y, part, mtiles, ntiles, C, ticket, fin0, work);
            hipLaunchKernelGGL(finalize, dim3((C + 31) / 32), dim3(1024), 0, 0, part, mtiles, C, fin0);
          } else if (var == 1) {
            hipLaunchKernelGGL(producer<1>, grid, dim3(T), 0, 0, y, part, mtiles, ntiles, C, ticket, fin1, work);
          } else if (var == 2) {
            hipLaunchKernelGGL(producer<2>, grid, dim3(T), 0, 0, y, part, mtiles, ntiles, C, ticket, fin1, work);
          } else if (var == 3) {
            hipLaunchKernelGGL(producer<3>, grid, dim3(T), 0, 0, y, part, mtiles, ntiles, C, ticket, fin1, work);
          } else {
            hipLaunchKernelGGL(producer<4>, grid, dim3(T), 0, 0, y, part, mtiles, ntiles, C, ticket, fin1, work);
          }
        }
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        us[var] = ms * 1e3f / reps;
      }
      // producer alone (no finalize at all), for reference
      for (int it = 0; it < reps + 20; ++it) {
        if (it == 20) CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(producer<0>, grid, dim3(T), 0, 0, y, part, mtiles, ntiles, C, ticket, fin0, work);
      }
      CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      const float alone = ms * 1e3f / reps;
      // correctness of the tail: one fresh launch each
      hipLaunchKernelGGL(producer<0>, grid, dim3(T), 0, 0, y, part, mtiles, ntiles, C, ticket, fin0, work);
      hipLaunchKernelGGL(finalize, dim3((C + 31) / 32), dim3(1024), 0, 0, part, mtiles, C, fin0);
      CK(hipMemset(fin1, 0, 2 * C * 4));
      hipLaunchKernelGGL(producer<1>, grid, dim3(T), 0, 0, y, part, mtiles, ntiles, C, ticket, fin1, work);
      std::vector<float> h0(2 * C), h1(2 * C);
      CK(hipMemcpy(h0.data(), fin0, 2 * C * 4, hipMemcpyDeviceToHost));
      CK(hipMemcpy(h1.data(), fin1, 2 * C * 4, hipMemcpyDeviceToHost));
      int bad = 0;
      for (int i = 0; i < 2 * C; ++i) bad += h0[i] != h1[i];
      // the no-fence tail under stress: 100 launches with different partials, each compared with the two-kernel result
      int bad3 = 0;
      for (int it = 0; it < 100; ++it) {
        const float seed = (float)(it * 3 + 1);
        hipLaunchKernelGGL(producer<0>, grid, dim3(T), 0, 0, y, part, mtiles, ntiles, C, ticket, fin0, work, seed);
        hipLaunchKernelGGL(finalize, dim3((C + 31) / 32), dim3(1024), 0, 0, part, mtiles, C, fin0);
        CK(hipMemcpy(h0.data(), fin0, 2 * C * 4, hipMemcpyDeviceToHost));
        CK(hipMemset(part, 0, (size_t)mtiles * 2 * C * 4));
        CK(hipMemset(fin1, 0, 2 * C * 4));
        hipLaunchKernelGGL(producer<3>, grid, dim3(T), 0, 0, y, part, mtiles, ntiles, C, ticket, fin1, work, seed);
        CK(hipMemcpy(h1.data(), fin1, 2 * C * 4, hipMemcpyDeviceToHost));
        for (int i = 0; i < 2 * C; ++i) bad3 += h0[i] != h1[i];
      }
      printf("   no fence (sc1 partial stores + vmcnt(0) + ticket): with tail %7.2f us | ticket only %7.2f us | tail sums differ over 100 launches: %d\n", us[3], us[4], bad3);
      printf("C=%4d tiles=%4d work=%5d | producer alone %7.2f us | + finalize kernel %7.2f | fence+ticket+tail %7.2f | fence+ticket only %7.2f | tail sums differ: %d\n",
             C, mtiles * ntiles, work, alone, us[0], us[1], us[2], bad);
      CK(hipFree(y)); CK(hipFree(part)); CK(hipFree(fin0)); CK(hipFree(fin1)); CK(hipFree(ticket));
    }
  }
  return 0;
}
