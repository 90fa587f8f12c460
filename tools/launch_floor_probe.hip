// Scratch probe: what a short dependent kernel costs on this GPU, back to back on one stream.
// Prints the steady-state time per launch (wall clock over 4000 launches) for empty kernels of several grids, for a
// kernel that makes ONE round trip to memory written by the previous launch, and for (large grid, small grid) pairs --
// the floor under the "main kernel + reduction" step of a small catalogue.
//     hipcc -O3 --offload-arch=gfx950 tools/launch_floor_probe.hip -o tools/launch_floor_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>

__global__ void empty_kernel() {}

// every thread reads what the previous launch's thread of another block wrote, adds, writes
__global__ void touch_kernel(const double* __restrict__ in, double* __restrict__ out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int j = (i + 4099) % n;
    out[i] = in[j] + 1.0;
}

// `loads` dependent-free 16-byte loads per thread (all in flight), summed, one store per thread
template <int LOADS>
__global__ void gather_kernel(const double2* __restrict__ in, double* __restrict__ out, int stride) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    double2 v[LOADS];
#pragma unroll
    for (int u = 0; u < LOADS; ++u) v[u] = in[i + (long)u * stride];
    double s = 0.0;
#pragma unroll
    for (int u = 0; u < LOADS; ++u) s += v[u].x + v[u].y;
    out[i] = s;
}

template <class F>
double per_launch_us(F&& launch, int n = 4000) {
    for (int i = 0; i < 200; ++i) launch(i);
    hipDeviceSynchronize();
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < n; ++i) launch(i);
    hipDeviceSynchronize();
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / n;
}

int main() {
    hipStream_t s;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    const int n = 2048 * 1024;
    double *a, *b;
    hipMalloc(&a, (size_t)n * 16 * 8);
    hipMalloc(&b, (size_t)n * 16 * 8);
    hipMemset(a, 0, (size_t)n * 16 * 8);
    hipMemset(b, 0, (size_t)n * 16 * 8);
    // clocks up
    per_launch_us([&](int) { touch_kernel<<<2048, 1024, 0, s>>>(a, b, n); }, 20000);
    const int grids[][2] = {{1, 64}, {1, 256}, {32, 64}, {32, 1024}, {256, 256}, {256, 1024}, {512, 512}, {1024, 256}, {1042, 256}, {2048, 256}};
    for (auto& g : grids)
        printf("empty  %5d x %4d : %6.2f us per launch\n", g[0], g[1],
               per_launch_us([&](int) { empty_kernel<<<g[0], g[1], 0, s>>>(); }));
    for (auto& g : grids)
        printf("touch  %5d x %4d : %6.2f us per launch (one round trip to the previous launch's output)\n", g[0], g[1],
               per_launch_us([&](int i) { touch_kernel<<<g[0], g[1], 0, s>>>(i & 1 ? a : b, i & 1 ? b : a, g[0] * g[1]); }));
    // pairs: big empty + small touch (the step of a small catalogue when the work itself is free)
    for (auto& g : grids) {
        if (g[0] < 256) continue;
        printf("pair   %5d x %4d empty + 32 x 1024 touch : %6.2f us per pair\n", g[0], g[1],
               per_launch_us([&](int i) {
                   empty_kernel<<<g[0], g[1], 0, s>>>();
                   touch_kernel<<<32, 1024, 0, s>>>(i & 1 ? a : b, i & 1 ? b : a, 32 * 1024);
               }));
    }
    // reduction-shaped reads: 32 blocks x 1024 threads, L loads of 16 B per thread in flight, input rewritten by a
    // 1024-block kernel in between (so it comes from memory, not from this XCD's L2)
    {
        const int stride = 32 * 1024;
#define GATHER(L)                                                                                                          \
    printf("writer 1024 x 256 + gather 32 x 1024 x %2d loads of 16 B (%5.0f KB): %6.2f us per pair\n", L,                  \
           32.0 * 1024 * L * 16 / 1024, per_launch_us([&](int) {                                                           \
               touch_kernel<<<1024, 256, 0, s>>>(b, a, 1024 * 256);                                                       \
               gather_kernel<L><<<32, 1024, 0, s>>>((const double2*)a, b, stride);                                         \
           }));
        GATHER(1) GATHER(2) GATHER(4) GATHER(8) GATHER(16)
#undef GATHER
#define GATHER2(G, T, L)                                                                                                   \
    printf("writer 1024 x 256 + gather %4d x %4d x %2d loads of 16 B (%5.0f KB): %6.2f us per pair\n", G, T, L,            \
           (double)G * T * L * 16 / 1024, per_launch_us([&](int) {                                                         \
               touch_kernel<<<1024, 256, 0, s>>>(b, a, 1024 * 256);                                                       \
               gather_kernel<L><<<G, T, 0, s>>>((const double2*)a, b, G * T);                                              \
           }));
        GATHER2(32, 64, 1) GATHER2(32, 64, 8) GATHER2(32, 256, 8) GATHER2(256, 256, 1) GATHER2(256, 256, 8) GATHER2(128, 256, 8)
        printf("writer 1024 x 256 alone: %6.2f us per launch\n",
               per_launch_us([&](int) { touch_kernel<<<1024, 256, 0, s>>>(b, a, 1024 * 256); }));
    }
    return 0;
}
