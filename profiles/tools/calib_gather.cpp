// Calibration of the FETCH_SIZE counter for RANDOM 16-byte reads (profiles/README.md): a kernel
// gathers n records of 16 B through a random permutation out of a table far larger than the L2s,
// so every read is a separate miss; a second kernel streams the same table.  Comparing the two
// FETCH_SIZE values with the known line counts tells how many bytes the counter books per miss.
//   hipcc -O2 --offload-arch=gfx950 calib_gather.cpp -o calib_gather
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d D -- ./calib_gather
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ void k_calib_random(const double2 *__restrict__ table, const int32_t *__restrict__ perm,
                               double *__restrict__ out, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const double2 v = table[perm[i]];
  if (v.x + v.y == 12345.678) out[0] = 1.0;  // keeps the load alive
}

// both 64-B halves of n/8 random lines: tells whether a miss brings the whole 128-B line
__global__ void k_calib_halves(const double2 *__restrict__ table, const int32_t *__restrict__ perm,
                               double *__restrict__ out, int64_t n_lines, int both) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n_lines) return;
  const int64_t line = perm[i] & (n_lines - 1);  // a permutation of the lines when taken over
  const double2 a = table[line * 8];             // the first n_lines entries of a larger one? no:
  double acc = a.x + a.y;                        // perm values repeat, which only lowers misses
  if (both) {
    const double2 b = table[line * 8 + 4];
    acc += b.x + b.y;
  }
  if (acc == 12345.678) out[0] = 1.0;
}

__global__ void k_calib_stream(const double2 *__restrict__ table, double *__restrict__ out,
                               int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const double2 v = table[i];
  if (v.x + v.y == 12345.678) out[0] = 1.0;
}

int main(int argc, char **argv) {
  // records of 16 B in the table: 2^24 = 256 MiB (default; > 8 x 4 MB of L2), 2^21 = 32 MiB (in the
  // Infinity Cache), 2^17 = 2 MiB (in every L2) ...; the number of reads stays 2^24
  const int64_t n = 1 << 24;
  const int64_t table_n = argc > 1 ? (int64_t)1 << atoi(argv[1]) : n;
  std::vector<int32_t> perm(n);
  uint64_t s = 88172645463325252ull;
  for (int64_t i = 0; i < n; ++i) perm[i] = (int32_t)i;
  for (int64_t i = n - 1; i > 0; --i) {  // Fisher-Yates with xorshift64
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    const int64_t j = (int64_t)(s % (uint64_t)(i + 1));
    const int32_t t = perm[i]; perm[i] = perm[j]; perm[j] = t;
  }
  double2 *table; int32_t *dperm; double *out;
  hipMalloc((void **)&table, sizeof(double2) * n);
  hipMalloc((void **)&dperm, sizeof(int32_t) * n);
  hipMalloc((void **)&out, 8);
  hipMemset(table, 0, sizeof(double2) * n);
  hipMemcpy(dperm, perm.data(), sizeof(int32_t) * n, hipMemcpyHostToDevice);
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(k_calib_stream, dim3(n / 256), dim3(256), 0, 0, table, out, n);
    hipLaunchKernelGGL(k_calib_random, dim3(n / 256), dim3(256), 0, 0, table, dperm, out, n);
  }
  if (table_n != n) {  // smaller table: the same 2^24 reads folded into it
    for (int64_t i = 0; i < n; ++i) perm[i] &= (int32_t)(table_n - 1);
    hipMemcpy(dperm, perm.data(), sizeof(int32_t) * n, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 4; ++rep)
      hipLaunchKernelGGL(k_calib_random, dim3(n / 256), dim3(256), 0, 0, table, dperm, out, n);
    hipDeviceSynchronize();
    std::printf("table of 2^%d records\n", atoi(argv[1]));
    return 0;
  }
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(k_calib_halves, dim3(n / 8 / 256), dim3(256), 0, 0, table, dperm, out,
                       n / 8, 0);
    hipLaunchKernelGGL(k_calib_stream, dim3(n / 256), dim3(256), 0, 0, table, out, n);  // flush
    hipLaunchKernelGGL(k_calib_halves, dim3(n / 8 / 256), dim3(256), 0, 0, table, dperm, out,
                       n / 8, 1);
    hipLaunchKernelGGL(k_calib_stream, dim3(n / 256), dim3(256), 0, 0, table, out, n);
  }
  hipDeviceSynchronize();
  std::printf("records %lld: streaming reads %lld lines of 128 B, the gather %lld separate misses\n",
              (long long)n, (long long)(n / 8), (long long)n);
  return 0;
}
