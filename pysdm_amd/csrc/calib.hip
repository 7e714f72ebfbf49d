// calib.hip -- the ceiling of the pair kernels' access pattern, measured where they run.
// The single-cell pair kernel (fused.hip: k_pair_all) spends its time in independent RANDOM
// 16-byte reads out of tables of 16 B per super-droplet (shuffle records, mirror records): each
// misses on one 64-B sector.  sdm_calib_random_sectors times exactly that pattern - n_reads
// 16-byte reads at hashed indices out of a table of table_records records - so that bench.py can
// quote the kernel against the random-sector rate of the same footprint next to the HBM peak
// (roofline.random_sector_ceiling_gbs / frac_of_ceiling).
#include "common.h"

#define TID_FLAT() ((int64_t)blockIdx.x * SDM_BLOCK + threadIdx.x)

// 64-bit mix (splitmix64 finaliser): read k goes to record mix(k) mod table_records
__device__ __forceinline__ uint64_t calib_mix(uint64_t x) {
  x += 0x9e3779b97f4a7c15ull;
  x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ull;
  x = (x ^ (x >> 27)) * 0x94d049bb133111ebull;
  return x ^ (x >> 31);
}

// record i = {i, 2 i + 1}
__global__ void __launch_bounds__(SDM_BLOCK)
k_calib_fill(ulonglong2 *__restrict__ table, int64_t n) {
  const int64_t i = TID_FLAT();
  if (i < n) table[i] = make_ulonglong2((unsigned long long)i, (unsigned long long)(2 * i + 1));
}

// four independent reads per thread (the pair kernels keep several gathers in flight per lane);
// the per-wave sum of what was read goes to `sums` so that the reads cannot be dropped and the
// caller can check that the kernel touched the records it claims
__global__ void __launch_bounds__(SDM_BLOCK)
k_calib_random(const ulonglong2 *__restrict__ table, int64_t table_records, int64_t n_reads,
               uint64_t salt, unsigned long long *__restrict__ sums) {
  const int64_t t = TID_FLAT();
  unsigned long long acc = 0;
  ulonglong2 v[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int64_t k = t * 4 + u;
    const uint64_t at = calib_mix((uint64_t)k ^ salt) % (uint64_t)table_records;
    v[u] = k < n_reads ? table[at] : make_ulonglong2(0, 0);
  }
#pragma unroll
  for (int u = 0; u < 4; ++u) acc += v[u].x + v[u].y;
  acc = (unsigned long long)wave_sum_i64((int64_t)acc);
  if (lane_id() == 0 && acc != 0)
    atomicAdd(&sums[(blockIdx.x & 63) * 16], acc);  // 64 slots, one cache line each
}

// the other side of the ledger when pricing a re-routing of the walks: what
// n scattered 8-byte stores cost - results delivered to random positions instead of read from them
__global__ void __launch_bounds__(SDM_BLOCK)
k_calib_random_write(unsigned long long *__restrict__ table, int64_t table_words, int64_t n_writes,
                     uint64_t salt) {
  const int64_t t = TID_FLAT();
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int64_t k = t * 4 + u;
    if (k < n_writes)
      table[calib_mix((uint64_t)k ^ salt) % (uint64_t)table_words] = (unsigned long long)k;
  }
}

extern "C" int sdm_calib_random_writes(sdm_ctx *ctx, int64_t table_words, int64_t n_writes,
                                       int repetitions, double *ms_per_launch) {
  ARG_TRY(ctx && table_words >= 1 && n_writes >= 1 && repetitions >= 1 && ms_per_launch);
  int rc = sdm_reserve(ctx, carve_size(sizeof(unsigned long long) * (size_t)table_words) + 512);
  if (rc) return rc;
  unsigned long long *table = (unsigned long long *)ctx->arena;
  hipStream_t s = ctx->stream;
  const dim3 grid(grid_for((n_writes + 3) / 4));
  hipEvent_t e0, e1;
  HIP_TRY(hipEventCreate(&e0));
  HIP_TRY(hipEventCreate(&e1));
  hipLaunchKernelGGL(k_calib_random_write, grid, dim3(SDM_BLOCK), 0, s, table, table_words,
                     n_writes, (uint64_t)0x5eed);
  HIP_TRY(hipEventRecord(e0, s));
  for (int r = 0; r < repetitions; ++r)
    hipLaunchKernelGGL(k_calib_random_write, grid, dim3(SDM_BLOCK), 0, s, table, table_words,
                       n_writes, (uint64_t)(r + 1) * 0x100000001b3ull);
  HIP_TRY(hipEventRecord(e1, s));
  LAUNCH_CHECK();
  HIP_TRY(hipEventSynchronize(e1));
  float ms = 0;
  HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
  HIP_TRY(hipEventDestroy(e0));
  HIP_TRY(hipEventDestroy(e1));
  *ms_per_launch = (double)ms / repetitions;
  return SDM_OK;
}

extern "C" int sdm_calib_random_sectors(sdm_ctx *ctx, int64_t table_records, int64_t n_reads,
                                        int repetitions, double *ms_per_launch,
                                        uint64_t *checksum) {
  ARG_TRY(ctx && table_records >= 1 && n_reads >= 1 && repetitions >= 1 && ms_per_launch &&
          checksum);
  const size_t table_bytes = sizeof(ulonglong2) * (size_t)table_records;
  const size_t sums_bytes = sizeof(unsigned long long) * 64 * 16;
  int rc = sdm_reserve(ctx, carve_size(table_bytes) + carve_size(sums_bytes) + 512);
  if (rc) return rc;
  Carver cv(ctx->arena);
  ulonglong2 *table = cv.take<ulonglong2>((size_t)table_records);
  unsigned long long *sums = cv.take<unsigned long long>(64 * 16);
  hipStream_t s = ctx->stream;
  hipLaunchKernelGGL(k_calib_fill, dim3(grid_for(table_records)), dim3(SDM_BLOCK), 0, s, table,
                     table_records);
  LAUNCH_CHECK();
  const dim3 grid(grid_for((n_reads + 3) / 4));
  hipEvent_t e0, e1;
  HIP_TRY(hipEventCreate(&e0));
  HIP_TRY(hipEventCreate(&e1));
  // warm-up launch (clocks, TLB), then the timed ones; every launch reads other records
  HIP_TRY(hipMemsetAsync(sums, 0, sums_bytes, s));
  hipLaunchKernelGGL(k_calib_random, grid, dim3(SDM_BLOCK), 0, s, table, table_records, n_reads,
                     (uint64_t)0x5eed, sums);
  HIP_TRY(hipMemsetAsync(sums, 0, sums_bytes, s));
  HIP_TRY(hipEventRecord(e0, s));
  for (int r = 0; r < repetitions; ++r)
    hipLaunchKernelGGL(k_calib_random, grid, dim3(SDM_BLOCK), 0, s, table, table_records, n_reads,
                       (uint64_t)(r + 1) * 0x100000001b3ull, sums);
  HIP_TRY(hipEventRecord(e1, s));
  LAUNCH_CHECK();
  HIP_TRY(hipEventSynchronize(e1));
  float ms = 0;
  HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
  HIP_TRY(hipEventDestroy(e0));
  HIP_TRY(hipEventDestroy(e1));
  unsigned long long host[64 * 16];
  HIP_TRY(hipMemcpyAsync(host, sums, sums_bytes, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  unsigned long long total = 0;
  for (int k = 0; k < 64; ++k) total += host[k * 16];
  *ms_per_launch = (double)ms / repetitions;
  *checksum = (uint64_t)total;
  return SDM_OK;
}
