// rider_probe.hip — how fast can Adam (zero gradient) stream p / m / v of the bio-synth tables (97 010 rows x 128 floats: 149 MB,
// inside the 256 MB Infinity Cache) in the forms the library uses?
//   (a) the rider loop of gqe_split.h (buffer-addressed wave blocks, U row slices in flight, no branch) at 4 / 8 / 16 waves per
//       workgroup, 1 / 2 / 4 rounds per wave;
//   (b) the pass's form: one float4 slice per thread, plain loads, one row per d/4 lanes.
//   hipcc --offload-arch=gfx950 -O3 -I graphqembed_amd/csrc tools/probes/rider_probe.hip -o /tmp/rider_probe && /tmp/rider_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#include "gqe_split.h"

__device__ unsigned long long g_first = ~0ull, g_last = 0ull;
template <int T>
__global__ __launch_bounds__(T) void k_rider(const GqeSplitRide r, int d) {
  if (threadIdx.x == 0) atomicMin(&g_first, (unsigned long long)wall_clock64());
  __shared__ int flag;
  split_rider<T / 64>(r, d, blockIdx.x, &flag);
  __syncthreads();
  if (threadIdx.x == 0) atomicMax(&g_last, (unsigned long long)wall_clock64());
}
__global__ __launch_bounds__(256) void k_left(const GqeSplitRide r, int d) { split_leftover(r, d, blockIdx.x * 4 + (threadIdx.x >> 6)); }
__global__ void k_book(const GqeSplitRide r, int done) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i == 0) *r.done = done;
  if (i < r.blocks * GQE_SPLIT_PWAVES) {
    const int j = i / GQE_SPLIT_PWAVES, w = i - j * GQE_SPLIT_PWAVES;
    int lo, hi;
    split_range(r, j, lo, hi);
    r.progress[i] = w < r.waves ? lo + w : 0x7fffffff;
  }
}
__global__ void k_fill(float* p, float* m, float* v, long long n) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) { p[i] = 1.f; m[i] = 0.5f; v[i] = 0.25f; }
}
__global__ void k_check(const float* p, const float* m, const float* v, long long n, float ep, float em, float ev, int* bad) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
    if (p[i] != ep || m[i] != em || v[i] != ev) atomicAdd(bad, 1);
}

__global__ __launch_bounds__(256) void k_plain(float* p, float* m, float* v, long long n4, float ss, float ibc, float b1c, float b2, float b2c, float eps) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    float4 pp = reinterpret_cast<float4*>(p)[i], mm = reinterpret_cast<float4*>(m)[i], vv = reinterpret_cast<float4*>(v)[i];
    gqe_adam1(pp.x, mm.x, vv.x, 0.f, ss, ibc, b1c, b2, b2c, eps);
    gqe_adam1(pp.y, mm.y, vv.y, 0.f, ss, ibc, b1c, b2, b2c, eps);
    gqe_adam1(pp.z, mm.z, vv.z, 0.f, ss, ibc, b1c, b2, b2c, eps);
    gqe_adam1(pp.w, mm.w, vv.w, 0.f, ss, ibc, b1c, b2, b2c, eps);
    reinterpret_cast<float4*>(m)[i] = mm;
    reinterpret_cast<float4*>(v)[i] = vv;
    reinterpret_cast<float4*>(p)[i] = pp;
  }
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main() {
  const int d = 128;
  const long long rows[5] = {10002, 10002, 27002, 40002, 10002};
  long long total = 0;
  for (long long r : rows) total += r;
  float *p, *m, *v;
  int32_t* stamp;
  CK(hipMalloc(&p, total * d * 4)); CK(hipMalloc(&m, total * d * 4)); CK(hipMalloc(&v, total * d * 4)); CK(hipMalloc(&stamp, total * 4));
  CK(hipMemset(p, 0, total * d * 4)); CK(hipMemset(m, 0, total * d * 4)); CK(hipMemset(v, 0, total * d * 4)); CK(hipMemset(stamp, 0, total * 4));
  GqeSplitRide r;
  memset(&r, 0, sizeof r);
  r.t.n = 5;
  long long off = 0;
  for (int k = 0; k < 5; ++k) {
    r.t.offset[k] = off * d; r.t.head_base[k] = off; r.t.rows[k] = rows[k];
    r.t.step_size[k] = 0.01f; r.t.bc2_sqrt[k] = 0.5f;
    r.t.blk_begin[k + 1] = r.t.blk_begin[k] + (int)((rows[k] + GQE_SPLIT_WROWS - 1) / GQE_SPLIT_WROWS);
    off += rows[k];
  }
  r.p = p; r.m = m; r.v = v; r.stamp = stamp; r.b1 = 0.9f; r.b2 = 0.999f; r.eps = 1e-8f;
  const int blocks_total = r.t.blk_begin[5];
  int32_t* book;
  CK(hipMalloc(&book, 4 * (256 + GQE_SPLIT_MAX_RIDERS * GQE_SPLIT_PWAVES)));
  r.done = book; r.progress = book + 256; r.tiles = 1;
  const double bytes = 2.0 * 3.0 * total * d * 4;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto time = [&](auto launch, const char* name) {
    for (int i = 0; i < 5; ++i) launch();
    hipEventRecord(e0, 0);
    const int reps = 50;
    for (int i = 0; i < reps; ++i) launch();
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-64s %7.2f us per launch  %6.2f TB/s (p, m, v in and out: %.0f MB)\n", name, ms * 1e3 / reps, bytes / (ms * 1e-3 / reps) / 1e12, bytes / 1e6);
  };
  auto config = [&](int riders, int waves) {
    r.blocks = riders; r.waves = waves; r.lead = riders;
    r.per = (blocks_total + riders - 1) / riders; r.share = 1; r.epoch = 1;
  };
  auto book_keep = [&](int done) { hipLaunchKernelGGL(k_book, dim3((r.blocks * GQE_SPLIT_PWAVES + 255) / 256 + 1), dim3(256), 0, 0, r, done); };
  auto leftover = [&]() { hipLaunchKernelGGL(k_left, dim3((r.blocks * r.waves + 3) / 4), dim3(256), 0, 0, r, d); };
  // correctness: every element takes exactly one step, whoever does it
  {
    float ep, em, ev;
    int* bad; CK(hipMalloc(&bad, 4));
    auto fill = [&]() { hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, p, m, v, total * d); };
    auto check = [&](const char* what) {
      int h = -1;
      hipMemset(bad, 0, 4);
      hipLaunchKernelGGL(k_check, dim3(4096), dim3(256), 0, 0, p, m, v, total * d, ep, em, ev, bad);
      hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost);
      printf("check %-52s: %d elements differ from one step\n", what, h);
    };
    config(128, 16); fill(); book_keep(0);
    hipLaunchKernelGGL(k_rider<1024>, dim3(128), dim3(1024), 0, 0, r, d);
    CK(hipMemcpy(&ep, p, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&em, m, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&ev, v, 4, hipMemcpyDeviceToHost));
    leftover();
    check("128 riders run to the end, leftover launch finds nothing");
    fill(); book_keep(1);   // tiles already through: the riders do nothing, the leftover launch everything
    hipLaunchKernelGGL(k_rider<1024>, dim3(128), dim3(1024), 0, 0, r, d);
    leftover();
    check("riders stop at once, leftover launch does everything");
    fill(); book_keep(0);   // only the first 40 riders ever run
    hipLaunchKernelGGL(k_rider<1024>, dim3(40), dim3(1024), 0, 0, r, d);
    leftover();
    check("40 of 128 riders run, leftover launch the other ranges");
    config(200, 8); fill(); book_keep(0);
    hipLaunchKernelGGL(k_rider<512>, dim3(77), dim3(512), 0, 0, r, d);
    leftover();
    check("8-wave riders: 77 of 200 run, leftover the rest");
  }
  char name[128];
  for (int riders : {48, 64, 96, 128, 256, 512}) {
    config(riders, 16);
    snprintf(name, sizeof name, "riders U=%d: %d x 16 waves run to the end", GQE_SPLIT_U, riders);
    time([&] { book_keep(0); hipLaunchKernelGGL(k_rider<1024>, dim3(riders), dim3(1024), 0, 0, r, d); }, name);
  }
  {   // how much of a rider launch's stream time lies outside its workgroups (dispatch, end-of-kernel write-back)?
    config(96, 16);
    book_keep(0);
    hipDeviceSynchronize();
    unsigned long long z0 = ~0ull, z1 = 0;
    hipMemcpyToSymbol(HIP_SYMBOL(g_first), &z0, 8);
    hipMemcpyToSymbol(HIP_SYMBOL(g_last), &z1, 8);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k_rider<1024>, dim3(96), dim3(1024), 0, 0, r, d);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    hipMemcpyFromSymbol(&z0, HIP_SYMBOL(g_first), 8);
    hipMemcpyFromSymbol(&z1, HIP_SYMBOL(g_last), 8);
    printf("one launch of 96 riders (aux %d): events %.2f us, first workgroup start -> last workgroup end %.2f us\n", GQE_SPLIT_AUX, ms * 1e3, (z1 - z0) / 100.0);
  }
  config(128, 16);
  time([&] { book_keep(1); leftover(); }, "leftover launch alone (128 x 16 pairs, 4-wave workgroups)");
  config(512, 16);
  time([&] { book_keep(1); leftover(); }, "leftover launch alone (512 x 16 pairs, 4-wave workgroups)");
  for (int wg : {2048, 8192, 32768, 97010 / 2}) {
    snprintf(name, sizeof name, "plain float4 slices, %d wgs of 256", wg);
    time([&] { hipLaunchKernelGGL(k_plain, dim3(wg), dim3(256), 0, 0, p, m, v, total * d / 4, 0.01f, 2.f, 0.1f, 0.999f, 0.001f, 1e-8f); }, name);
  }
  return 0;
}
