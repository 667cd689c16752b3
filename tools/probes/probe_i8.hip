// Probe: operand lane map of v_mfma_i32_32x32x32_i8 on gfx950 (cdna_hip_programming.md section 3:
// "Other dtypes: check the map with exact integer data before relying on it").
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
using i32x4 = __attribute__((ext_vector_type(4))) int;
using i32x16 = __attribute__((ext_vector_type(16))) int;

// candidate 0: lane (r = l&31, h = l>>5) holds k = 16h + j, j = 0..15
// candidate 1: bytes 0-7: k = 8h + j ; bytes 8-15: k = 16 + 8h + (j-8)
__global__ void probe(const signed char *A, const signed char *B, int *C, int cand) {
  const int l = threadIdx.x, r = l & 31, h = l >> 5;
  signed char a[16], b[16];
  for (int j = 0; j < 16; ++j) {
    const int k = cand == 0 ? 16 * h + j : (j < 8 ? 8 * h + j : 16 + 8 * h + (j - 8));
    a[j] = A[r * 32 + k];   // A[row r][k]
    b[j] = B[k * 32 + r];   // B[k][col r]
  }
  i32x4 av, bv;
  for (int q = 0; q < 4; ++q) {
    av[q] = (a[4 * q] & 255) | ((a[4 * q + 1] & 255) << 8) | ((a[4 * q + 2] & 255) << 16) | ((a[4 * q + 3] & 255) << 24);
    bv[q] = (b[4 * q] & 255) | ((b[4 * q + 1] & 255) << 8) | ((b[4 * q + 2] & 255) << 16) | ((b[4 * q + 3] & 255) << 24);
  }
  i32x16 acc = {0};
  acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(av, bv, acc, 0, 0, 0);
  for (int q = 0; q < 16; ++q) {
    const int row = (q & 3) + 8 * (q >> 2) + 4 * h, col = r;
    C[row * 32 + col] = acc[q];
  }
}

int main() {
  std::vector<signed char> A(1024), B(1024);
  srand(1);
  for (auto &v : A) v = (signed char)(rand() % 255 - 127);
  for (auto &v : B) v = (signed char)(rand() % 15 - 8);     // asymmetric
  std::vector<int> ref(1024, 0);
  for (int i = 0; i < 32; ++i)
    for (int j = 0; j < 32; ++j) {
      int s = 0;
      for (int k = 0; k < 32; ++k) s += (int)A[i * 32 + k] * (int)B[k * 32 + j];
      ref[i * 32 + j] = s;
    }
  signed char *dA, *dB;
  int *dC;
  hipMalloc(&dA, 1024); hipMalloc(&dB, 1024); hipMalloc(&dC, 4096);
  hipMemcpy(dA, A.data(), 1024, hipMemcpyHostToDevice);
  hipMemcpy(dB, B.data(), 1024, hipMemcpyHostToDevice);
  for (int cand = 0; cand < 2; ++cand) {
    hipMemset(dC, 0, 4096);
    probe<<<1, 64>>>(dA, dB, dC, cand);
    std::vector<int> out(1024);
    hipMemcpy(out.data(), dC, 4096, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 1024; ++i) bad += out[i] != ref[i];
    printf("candidate %d: %d mismatches of 1024\n", cand, bad);
  }
  return 0;
}
