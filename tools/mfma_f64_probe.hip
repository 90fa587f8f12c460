// Scratch probe: can the f64 matrix pipe take the bilinear part of a star-walker term off the f64 VALU?
//   (a) v_mfma_f64_16x16x4_f64 alone, (b) an independent v_fma_f64 stream alone, (c) both interleaved in one wave at the
//   ratio the BGFIXED kernel would have (1 MFMA per NFMA VALU instructions), on a fully occupied chip.
// Prints ns per instruction-slot per SIMD; (c) tells what one MFMA costs the VALU stream.
//     hipcc --offload-arch=gfx950 -O3 tools/mfma_f64_probe.hip -o tools/mfma_f64_probe && tools/mfma_f64_probe
#include <hip/hip_runtime.h>
#include <cstdio>

typedef double double4_t __attribute__((ext_vector_type(4)));

template <int MODE, int NFMA>
__global__ __launch_bounds__(256) void k(double* out, int iters, double seed) {
    double a[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = seed + threadIdx.x * 1e-3 + j;
    double4_t acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
    const double ma = seed + (threadIdx.x & 15), mb = seed * 0.5 + (threadIdx.x >> 4);
    for (int i = 0; i < iters; ++i) {
        if (MODE != 1) {
            acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(ma, mb, acc0, 0, 0, 0);
        }
        if (MODE != 0) {
#pragma unroll
            for (int r = 0; r < NFMA / 8; ++r)
#pragma unroll
                for (int j = 0; j < 8; ++j) a[j] = __builtin_fma(a[j], 1.0000001, 1e-9);
        }
        if (MODE == 3) {      // second independent MFMA per group
            acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(mb, ma, acc1, 0, 0, 0);
        }
    }
    double s = acc0[0] + acc0[1] + acc0[2] + acc0[3] + acc1[0] + acc1[1] + acc1[2] + acc1[3];
#pragma unroll
    for (int j = 0; j < 8; ++j) s += a[j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE, int NFMA> void run(const char* name, double* d, int blocks_per_cu) {
    const int iters = 4000, grid = 256 * blocks_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE, NFMA><<<grid, 256>>>(d, 100, 1.5);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE, NFMA><<<grid, 256>>>(d, iters, 1.5);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double groups = (double)grid * 4 / 1024 * iters;          // loop iterations per SIMD
    const double ns_group = ms * 1e6 / groups;
    const int n_mfma = MODE == 1 ? 0 : (MODE == 3 ? 2 : 1), n_fma = MODE == 0 ? 0 : NFMA;
    printf("%-44s waves/SIMD %d  %8.3f ms  %8.2f ns per group (%d MFMA + %d FMA)", name, blocks_per_cu, ms, ns_group, n_mfma, n_fma);
    if (n_fma) printf("  = %6.3f ns per FMA slot", ns_group / n_fma);
    printf("\n");
}

int main() {
    double* d; hipMalloc(&d, 256 * 8 * 256 * 8);
    for (int occ : {8, 4}) {
        if (occ == 8) {
            run<0, 0>("mfma_f64_16x16x4 alone", d, 8);
            run<1, 64>("64 v_fma_f64 alone", d, 8);
            run<2, 64>("1 MFMA + 64 FMA", d, 8);
            run<2, 32>("1 MFMA + 32 FMA", d, 8);
            run<2, 16>("1 MFMA + 16 FMA", d, 8);
            run<3, 64>("2 MFMA + 64 FMA", d, 8);
            run<3, 32>("2 MFMA + 32 FMA", d, 8);
        } else {
            run<0, 0>("mfma_f64_16x16x4 alone", d, 4);
            run<1, 64>("64 v_fma_f64 alone", d, 4);
            run<2, 64>("1 MFMA + 64 FMA", d, 4);
            run<3, 64>("2 MFMA + 64 FMA", d, 4);
        }
    }
    return 0;
}
