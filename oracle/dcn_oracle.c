/*
 * oracle/dcn_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU oracle for the deformable-convolution hot path of Zhen-Dong/CoDeNet: a restatement
 * (see dcn_oracle_impl.h for the per-function reference citations) of
 *   lib/models/external/src/dcn_deform_conv_cuda_kernel.cu   (device arithmetic)
 *   lib/models/external/src/dcn_deform_conv_cuda.cpp         (host composition)
 * in fp32 and fp64.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may load this library; the product path (codenet_amd/) never does.
 *
 * PARITY PINNING: the reference ships no golden vectors / known-answer tests for this
 * extension and its native code cannot be built here (CUDA + THC), so
 * this oracle is pinned by (a) independent PyTorch primitives (F.grid_sample, F.conv2d,
 * fp64 autograd) in tests/test_oracle.py and (b) the portable known-answer properties of
 * lib/models/networks/DCNv2/test.py:32-95.  "parity unpinned" by reference-owned vectors.
 *
 * Build: see oracle/Makefile  (gcc -O2 -fopenmp -ffp-contract=off -shared -fPIC).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define T float
#define FN(name) name##_f32
#include "dcn_oracle_impl.h"
#undef T
#undef FN

#define T double
#define FN(name) name##_f64
#include "dcn_oracle_impl.h"
#undef T
#undef FN

int dcn_oracle_abi_version(void) { return 1; }

/* OpenMP team size of the loops above (bench.py's single-thread CPU baseline leg); returns the old value. */
#ifdef _OPENMP
#include <omp.h>
int dcn_oracle_set_threads(int n) {
  const int old = omp_get_max_threads();
  if (n > 0) omp_set_num_threads(n);
  return old;
}
#else
int dcn_oracle_set_threads(int n) { (void)n; return 1; }
#endif
