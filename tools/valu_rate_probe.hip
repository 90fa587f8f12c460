// Scratch probe: issue cost of the f64 VALU instructions the kernels use, on a fully occupied MI355X.
// Prints cycles per wave-instruction per SIMD, assuming the clock reported by the runtime.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define OPS 8
template <int OP>
__global__ __launch_bounds__(256) void k(double* out, int iters, double seed) {
    double a[OPS];
    int ei[OPS];
#pragma unroll
    for (int j = 0; j < OPS; ++j) { a[j] = seed + threadIdx.x * 1e-3 + j; ei[j] = 0; }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < OPS; ++j) {
            if (OP == 0) a[j] = __builtin_fma(a[j], 1.0000001, 1e-9);
            if (OP == 1) a[j] = a[j] * 1.0000001;
            if (OP == 2) a[j] = a[j] + 1e-9;
            if (OP == 3) a[j] = __builtin_amdgcn_rsq(a[j]) + 2.0;
            if (OP == 4) { int e; a[j] = __builtin_frexp(a[j], &e) + 1.0; ei[j] += e; }
            if (OP == 5) a[j] = __builtin_ldexp(a[j], (i & 1) ? 1 : -1);
            if (OP == 6) a[j] = __builtin_rint(a[j] * 1.5) ;
            if (OP == 7) { ei[j] += (int)a[j]; a[j] += 1e-9; }
            if (OP == 8) a[j] = __builtin_amdgcn_rcp(a[j]) + 1.0;
            if (OP == 9) a[j] = __builtin_fmax(a[j], 1.5) * 1.0000001;
            if (OP == 10) a[j] = __builtin_amdgcn_frexp_mant(a[j]) + 1.0;
            if (OP == 11) ei[j] += __builtin_amdgcn_frexp_exp(a[j] + (double)i);
        }
    }
    double s = 0; int t = 0;
#pragma unroll
    for (int j = 0; j < OPS; ++j) { s += a[j]; t += ei[j]; }
    out[blockIdx.x * 256 + threadIdx.x] = s + t;
}

// float32 counterparts (round 3: the roof of the MCD_F32 / MCD_F32_ACC64 kernels)
typedef float float2_t __attribute__((ext_vector_type(2)));
template <int OP>
__global__ __launch_bounds__(256) void kf(double* out, int iters, float seed) {
    float a[OPS];
    float2_t p[OPS];
    double d[OPS];
    int ei[OPS];
#pragma unroll
    for (int j = 0; j < OPS; ++j) { a[j] = seed + threadIdx.x * 1e-3f + j; p[j] = float2_t{a[j], a[j] + 0.5f}; d[j] = a[j]; ei[j] = 0; }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < OPS; ++j) {
            if (OP == 0) a[j] = __builtin_fmaf(a[j], 1.0000001f, 1e-9f);
            if (OP == 1) a[j] = a[j] * 1.0000001f;
            if (OP == 2) a[j] = a[j] + 1e-9f;
            if (OP == 3) a[j] = __builtin_amdgcn_rsqf(a[j]) + 2.0f;
            if (OP == 4) a[j] = __builtin_amdgcn_rcpf(a[j]) + 1.0f;
            if (OP == 5) a[j] = __builtin_amdgcn_exp2f(a[j] * 0.01f) + 0.5f;
            if (OP == 6) a[j] = __builtin_amdgcn_logf(a[j]) + 3.0f;
            if (OP == 7) { int e; a[j] = __builtin_frexpf(a[j], &e) + 1.0f; ei[j] += e; }
            if (OP == 8) p[j] = __builtin_elementwise_fma(p[j], float2_t{1.0000001f, 0.9999999f}, float2_t{1e-9f, 1e-9f});   // v_pk_fma_f32
            if (OP == 9) d[j] += (double)a[j], a[j] = a[j] * 1.0000001f;               // f32 mul + cvt + f64 add (the f32acc64 sums)
            if (OP == 10) a[j] = __builtin_fmaxf(a[j], 1.5f) * 1.0000001f;
        }
    }
    double s = 0; int t = 0;
#pragma unroll
    for (int j = 0; j < OPS; ++j) { s += a[j] + p[j].x + p[j].y + d[j]; t += ei[j]; }
    out[blockIdx.x * 256 + threadIdx.x] = s + t;
}

template <int OP> double runf(const char* name, double ops_per, double* d) {
    const int iters = 20000, grid = 256 * 8;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    kf<OP><<<grid, 256>>>(d, 100, 1.5f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    kf<OP><<<grid, 256>>>(d, iters, 1.5f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double winstr = (double)grid * 4 / 1024 * iters * OPS;
    double ns_per = ms * 1e6 / winstr;
    printf("%-36s %8.3f ms   %6.2f ns per loop slot (= %5.1f cycles @2.4GHz), %.0f wave-instruction(s) per slot\n", name, ms, ns_per, ns_per * 2.4, ops_per);
    return ns_per;
}

template <int OP> double run(const char* name, double extra_ops_per, double* d) {
    const int iters = 20000, grid = 256 * 8;   // 8 blocks of 4 waves per CU: 8 waves per SIMD
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<OP><<<grid, 256>>>(d, 100, 1.5);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<OP><<<grid, 256>>>(d, iters, 1.5);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // wave-instructions per SIMD: grid*4 waves / 1024 SIMDs * iters * OPS
    double winstr = (double)grid * 4 / 1024 * iters * OPS;
    double ns_per = ms * 1e6 / winstr;
    printf("%-28s %8.3f ms   %6.2f ns per wave-instr-slot (= %5.1f cycles @2.4GHz) incl. %.0f helper op(s)\n", name, ms, ns_per, ns_per * 2.4, extra_ops_per);
    return ns_per;
}

int main() {
    double* d; hipMalloc(&d, 256 * 8 * 256 * 8);
    run<0>("v_fma_f64", 0, d);
    run<1>("v_mul_f64", 0, d);
    run<2>("v_add_f64", 0, d);
    run<3>("v_rsq_f64 + add", 1, d);
    run<8>("v_rcp_f64 + add", 1, d);
    run<4>("frexp mant+exp + add + iadd", 3, d);
    run<10>("v_frexp_mant_f64 + add", 1, d);
    run<11>("v_frexp_exp_i32_f64 + add + cvt", 2, d);
    run<5>("v_ldexp_f64", 0, d);
    run<6>("v_rndne_f64 + mul", 1, d);
    run<7>("v_cvt_i32_f64 + add + iadd", 2, d);
    run<9>("v_max_f64 + mul", 1, d);
    printf("-- float32 --\n");
    runf<0>("v_fma_f32", 1, d);
    runf<1>("v_mul_f32", 1, d);
    runf<2>("v_add_f32", 1, d);
    runf<8>("v_pk_fma_f32 (2 floats per lane)", 1, d);
    runf<3>("v_rsq_f32 + add", 2, d);
    runf<4>("v_rcp_f32 + add", 2, d);
    runf<5>("v_exp_f32 + mul + add", 3, d);
    runf<6>("v_log_f32 + add", 2, d);
    runf<7>("frexp f32 mant+exp + add + iadd", 4, d);
    runf<9>("f32 mul + v_cvt_f64_f32 + v_add_f64", 3, d);
    runf<10>("v_max_f32 + mul", 2, d);
    return 0;
}
