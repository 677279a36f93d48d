// Discovers the operand layout of v_smfmac_i32_16x16x128_i8 (gfx950) by one-hot probing; prints the maps as JSON.
//   pass B: A = all ones (every kept slot), idx = 0xEEEE.. (keeps positions 2, 3? no: see below) ... one-hot B at (lane tl, byte tb):
//           which D (row, col) entries light up tells the column of B lane tl and whether byte tb is at a kept K position.
//   pass A: B[k][col] = k + 1 for every col (the layout of B found in pass B is not needed: we build B through the dense formula once
//           pass B has told us (lane, byte) -> k); one-hot A at (lane al, slot as) with a given idx word: D = k + 1 of the dense
//           position that slot multiplies.
// Each wave of the launch is one test case; results are analysed on the host.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
typedef int i4 __attribute__((ext_vector_type(4)));
typedef int i8v __attribute__((ext_vector_type(8)));

// generic: every wave gets its own A (16 B / lane), B (32 B / lane), idx (4 B / lane) from memory
__global__ void k_cases(const i4* a, const i8v* b, const int* idx, i4* d) {
  const size_t w = blockIdx.x, l = threadIdx.x;
  i4 acc = {0, 0, 0, 0};
  acc = __builtin_amdgcn_smfmac_i32_16x16x128_i8(a[w * 64 + l], b[w * 64 + l], acc, idx[w * 64 + l], 0, 0);
  d[w * 64 + l] = acc;
}

int main() {
  // ---------------- pass B: one-hot B, A all ones, idx all 0x44444444 (first kept = position 0, second = position 1) -------------
  // D[row][col] = sum over kept dense k of B[k][col]; with one-hot B (value 1) a full column `col` of D shows 1 iff the byte sits at a kept k
  const int NB = 64 * 32;
  std::vector<int8_t> ha((size_t)NB * 64 * 16, 1), hb((size_t)NB * 64 * 32, 0);
  std::vector<int> hi((size_t)NB * 64, 0x44444444);
  for (int t = 0; t < NB; ++t) hb[((size_t)t * 64 + t / 32) * 32 + t % 32] = 1;
  i4 *da, *dd; i8v* db; int* di;
  CK(hipMalloc(&da, ha.size())); CK(hipMalloc(&db, hb.size())); CK(hipMalloc(&di, hi.size() * 4)); CK(hipMalloc(&dd, (size_t)NB * 64 * 16));
  CK(hipMemcpy(da, ha.data(), ha.size(), hipMemcpyHostToDevice)); CK(hipMemcpy(db, hb.data(), hb.size(), hipMemcpyHostToDevice));
  CK(hipMemcpy(di, hi.data(), hi.size() * 4, hipMemcpyHostToDevice));
  k_cases<<<NB, 64>>>(da, db, di, dd);
  std::vector<int> hd((size_t)NB * 256);
  CK(hipMemcpy(hd.data(), dd, hd.size() * 4, hipMemcpyDeviceToHost));
  printf("{\"pass\":\"B one-hot, idx nibble 0x4 (positions 0,1 kept)\",\"per_lane_byte\":[");
  for (int t = 0; t < NB; ++t) {
    // which columns (D lane & 15) are lit, and in how many (row) entries
    int colmask = 0, cnt = 0;
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) if (hd[(size_t)t * 256 + l * 4 + r]) { colmask |= 1 << (l & 15); ++cnt; }
    int col = -1;
    for (int c = 0; c < 16; ++c) if (colmask == (1 << c)) col = c;
    if (t % 32 == 0) printf("%s\n  {\"lane\":%d,\"col_hit_per_byte\":[", t ? "," : "", t / 32);
    printf("%s[%d,%d]", t % 32 ? "," : "", col, cnt);
    if (t % 32 == 31) printf("]}");
  }
  printf("]}\n");

  // ---------------- pass A: B holds k + 1 at the positions the straightforward layout assumes; one-hot A --------------------------------
  // B built per candidate layout is not known yet, so instead use B = all ones except one dense "marker": simpler: B[lane][byte] = byte + 32 * (lane >> 4) + 1
  // (i.e. if the assumed layout K = 32 g + byte were right, D would read k + 1).  The values that come back are (byte index within the
  // lane group g) + 32 g + 1 of whatever B bytes the kept slot multiplies — combined with pass B this pins the K map.
  const unsigned idxs[6] = {0x44444444u, 0x88888888u, 0xCCCCCCCCu, 0x99999999u, 0xDDDDDDDDu, 0xEEEEEEEEu};   // (p0,p1) = (0,1),(0,2),(0,3),(1,2),(1,3),(2,3)
  const int NA = 6 * 64 * 16;
  std::vector<int8_t> ha2((size_t)NA * 64 * 16, 0), hb2((size_t)NA * 64 * 32);
  std::vector<int> hi2((size_t)NA * 64);
  for (int t = 0; t < NA; ++t) {
    const int e = t / (64 * 16), al = (t / 16) % 64, as = t % 16;
    ha2[((size_t)t * 64 + al) * 16 + as] = 1;
    for (int l = 0; l < 64; ++l) {
      hi2[(size_t)t * 64 + l] = (int)idxs[e];
      for (int p = 0; p < 32; ++p) hb2[((size_t)t * 64 + l) * 32 + p] = (int8_t)(p + 32 * (l >> 4) + 1 - 128 * ((p + 32 * (l >> 4) + 1) > 127));
    }
  }
  i4 *da2, *dd2; i8v* db2; int* di2;
  CK(hipMalloc(&da2, ha2.size())); CK(hipMalloc(&db2, hb2.size())); CK(hipMalloc(&di2, hi2.size() * 4)); CK(hipMalloc(&dd2, (size_t)NA * 64 * 16));
  CK(hipMemcpy(da2, ha2.data(), ha2.size(), hipMemcpyHostToDevice)); CK(hipMemcpy(db2, hb2.data(), hb2.size(), hipMemcpyHostToDevice));
  CK(hipMemcpy(di2, hi2.data(), hi2.size() * 4, hipMemcpyHostToDevice));
  k_cases<<<NA, 64>>>(da2, db2, di2, dd2);
  std::vector<int> hd2((size_t)NA * 256);
  CK(hipMemcpy(hd2.data(), dd2, hd2.size() * 4, hipMemcpyDeviceToHost));
  for (int e = 0; e < 6; ++e) {
    printf("{\"pass\":\"A one-hot\",\"idx\":\"0x%08X\",\"per_lane\":[", idxs[e]);
    for (int al = 0; al < 64; ++al) {
      printf("%s\n  {\"lane\":%d,\"slot_to_[row,value]\":[", al ? "," : "", al);
      for (int as = 0; as < 16; ++as) {
        const int t = (e * 64 + al) * 16 + as;
        int row = -1, val = 0, nrows = 0;
        for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) { const int v = hd2[(size_t)t * 256 + l * 4 + r]; if (v && (l & 15) == 0) { row = 4 * (l >> 4) + r; val = v; ++nrows; } }
        printf("%s[%d,%d%s]", as ? "," : "", row, val, nrows > 1 ? ",\"multi\"" : "");
      }
      printf("]}");
      if (al == 17) { printf(",\n  \"...\""); al = 47; }   // lanes 0..17 and 48..63 are enough to see the pattern
    }
    printf("]}\n");
  }
  return 0;
}
