// FETCH_SIZE calibration for the access patterns of this library's per-observation kernels:
// MI355X_MICROARCH.md calibrates rocprofv3's FETCH_SIZE only for wide coalesced streams (16 B per lane: the counter reports
// HALF the bytes); the gathers here are 24-, 48-, 64- and 192-byte records at random addresses.  This program reads N records of
// R bytes, each exactly once, from a table far larger than the 256 MiB Infinity Cache (so that every record is an HBM fetch),
// one record per lane, and prints the bytes it asked for; tools/fetch_calib.sh runs it under `rocprofv3 --pmc FETCH_SIZE` and
// divides.  A 16-byte-per-lane streaming read of the same table is the control (expected ratio 0.5).
//   usage: fetch_calib <record bytes: 8|16|24|48|64|192|0 = stream> [table MiB = 4096] [records = 8 M]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// record i of the walk -> record slot: an odd multiplier modulo a power of two is a permutation of the slots
__device__ __forceinline__ uint64_t slot_of(uint64_t i, uint64_t mask) { return (i * 0x9E3779B97F4A7C15ull + 0x632BE59BD9B4E019ull) & mask; }

template <int R> __global__ void k_gather(const char *__restrict__ table, uint64_t mask, uint64_t n, double *__restrict__ out) {
  double acc = 0;
  for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const double *p = reinterpret_cast<const double *>(table + slot_of(i, mask) * (uint64_t)R);
#pragma unroll
    for (int k = 0; k < R / 8; ++k) acc += p[k];
  }
  if (acc == 1.2345e300) out[0] = acc; // keeps the loads
}
__global__ void k_stream(const double2 *__restrict__ table, uint64_t n16, double *__restrict__ out) {
  double acc = 0;
  for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * blockDim.x) { const double2 v = table[i]; acc += v.x + v.y; }
  if (acc == 1.2345e300) out[0] = acc;
}

int main(int argc, char **argv) {
  const int R = argc > 1 ? std::atoi(argv[1]) : 64;
  const size_t mib = argc > 2 ? std::strtoull(argv[2], nullptr, 10) : 4096;
  uint64_t n = argc > 3 ? std::strtoull(argv[3], nullptr, 10) : (8ull << 20);
  const size_t bytes = mib << 20;
  char *table = nullptr;
  double *out = nullptr;
  CHECK(hipMalloc(reinterpret_cast<void **>(&table), bytes));
  CHECK(hipMalloc(reinterpret_cast<void **>(&out), 64));
  CHECK(hipMemset(table, 1, bytes));
  CHECK(hipDeviceSynchronize());
  double asked = 0;
  if (R == 0) {
    const uint64_t n16 = bytes / 16;
    k_stream<<<256 * 8, 256>>>(reinterpret_cast<const double2 *>(table), n16, out);
    asked = (double)bytes;
    n = n16;
  } else {
    uint64_t slots = 1;
    while (slots * 2 * (uint64_t)R <= bytes) slots *= 2; // power of two so that the multiplicative walk is a permutation
    if (n > slots) n = slots;
    const uint64_t mask = slots - 1;
    switch (R) {
    case 8: k_gather<8><<<256 * 8, 256>>>(table, mask, n, out); break;
    case 16: k_gather<16><<<256 * 8, 256>>>(table, mask, n, out); break;
    case 24: k_gather<24><<<256 * 8, 256>>>(table, mask, n, out); break;
    case 48: k_gather<48><<<256 * 8, 256>>>(table, mask, n, out); break;
    case 64: k_gather<64><<<256 * 8, 256>>>(table, mask, n, out); break;
    case 192: k_gather<192><<<256 * 8, 256>>>(table, mask, n, out); break;
    default: std::fprintf(stderr, "record bytes: 8 16 24 48 64 192 or 0\n"); return 2;
    }
    asked = (double)n * R;
  }
  CHECK(hipGetLastError());
  CHECK(hipDeviceSynchronize());
  std::printf("CALIB record_bytes %d records %llu asked_bytes %.0f table_MiB %zu\n", R, (unsigned long long)n, asked, mib);
  return 0;
}
