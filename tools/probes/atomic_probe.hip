// atomic_probe.hip — how fast do float atomics on ONE 256-byte row go (gfx950)?  The hot-row accumulators (GqeHot) take
// thousands of wave-wide atomic rows per step on the same addresses; this measures the cost per atomic row
//   (a) from workgroups of ONE XCD, (b) from all eight, (c) spread over 8 per-XCD replicas 16 KB apart, (d) over 8 / 32 replicas
//   that sit NEXT to each other (the layout hot_acc uses), (e) LDS ds_add_f32 for comparison.
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/probes/atomic_probe.hip -o /tmp/atomic_probe && /tmp/atomic_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void k_atomic(float* acc, int per_wave, int mode, int floats) {
  int xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  xcc &= 7;
  if (mode == 0 && xcc != 0) return;                       // one XCD only
  float* p = acc;
  if (mode == 2) p += (size_t)xcc * 4096;                                         // per-XCD replica, 16 KB apart
  if (mode == 3) p += (size_t)xcc * floats;                                       // 8 adjacent replicas
  if (mode == 4) p += (size_t)(xcc * 4 + ((threadIdx.x >> 6) & 3)) * floats;      // 32 adjacent replicas: (XCD, wave mod 4)
  const int lane = threadIdx.x & 63;
  for (int i = 0; i < per_wave; ++i)
    for (int c = 0; c < floats / 64; ++c) unsafeAtomicAdd(p + lane + 64 * c, 1.0f);
}

__global__ void k_lds(float* out, int per_wave, int floats) {
  __shared__ float s[1024];
  for (int i = threadIdx.x; i < 1024; i += blockDim.x) s[i] = 0.f;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  for (int i = 0; i < per_wave; ++i)
    for (int c = 0; c < floats / 64; ++c) unsafeAtomicAdd(s + lane + 64 * c, 1.0f);
  __syncthreads();
  if (threadIdx.x < 64) out[blockIdx.x * 64 + threadIdx.x] = s[threadIdx.x];
}

int main() {
  float* acc;
  hipMalloc(&acc, 1 << 20);
  hipMemset(acc, 0, 1 << 20);
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  const int blocks = 256, threads = 1024, per_wave = 64;   // 256 x 16 waves x 64 rows = 262144 atomic rows
  for (int floats : {64, 256}) {
    for (int mode = 0; mode < 5; ++mode) {
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(a);
        hipLaunchKernelGGL(k_atomic, dim3(blocks), dim3(threads), 0, 0, acc, per_wave, mode, floats);
        hipEventRecord(b);
        hipEventSynchronize(b);
      }
      float ms;
      hipEventElapsedTime(&ms, a, b);
      const double rows = (double)blocks * 16 * per_wave / (mode == 0 ? 8 : 1);
      const char* names[5] = {"ONE XCD, one target", "8 XCDs, one target", "8 XCDs, 8 replicas 16 KB apart", "8 XCDs, 8 adjacent replicas",
                              "8 XCDs, 32 adjacent replicas"};
      printf("global rows of %d floats, %s: %.1f us for %.0f rows -> %.1f ns per row overall\n", floats, names[mode], ms * 1e3, rows,
             ms * 1e6 / rows);
    }
  }
  float* out;
  hipMalloc(&out, blocks * 64 * 4);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(a);
    hipLaunchKernelGGL(k_lds, dim3(blocks), dim3(threads), 0, 0, out, per_wave, 256);
    hipEventRecord(b);
    hipEventSynchronize(b);
  }
  float ms;
  hipEventElapsedTime(&ms, a, b);
  printf("LDS rows of 256 floats (16 waves x %d rows per workgroup on one row): %.1f us -> %.1f ns per row per CU\n", per_wave, ms * 1e3,
         ms * 1e6 / (16.0 * per_wave));
  return 0;
}
