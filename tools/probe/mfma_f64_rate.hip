// Probe (round 4): is the FP64 matrix pipe worth anything for the Gaussian quadratic form on gfx950?
//   hipcc --offload-arch=gfx950 -O3 tools/probe/mfma_f64_rate.hip -o build_ab/mfma_f64_rate && build_ab/mfma_f64_rate
// Shader cycles (s_memtime) per instruction of ONE wave on one SIMD: v_mfma_f64_16x16x4_f64 with 4 independent accumulators
// (issue rate) and on one accumulator (dependent latency), and v_fma_f64 with 8 independent chains (issue rate).
// The dense likelihood at D = 32 is, per 64-walker tile: symmetric VALU form 560 v_fma_f64 per lane; full matrix form
// Y = Q A on the matrix pipe 64 walkers x 32 x 32 = 64 MFMAs of 16 x 16 x 4 (+ the row dot).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void k_rate(double* out, long long* cyc, int n) {
    const int l = threadIdx.x;
    double a = 1.0 + l * 1e-3, b = 1.0 - l * 1e-3;
    d4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i) {
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
    }
    asm volatile("s_nop 0" :: "v"(c0), "v"(c1), "v"(c2), "v"(c3));
    long long t1 = __builtin_amdgcn_s_memtime();
    d4 d0 = {0, 0, 0, 0};
    for (int i = 0; i < 4 * n; ++i) d0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, d0, 0, 0, 0);
    asm volatile("s_nop 0" :: "v"(d0));
    long long t2 = __builtin_amdgcn_s_memtime();
    double f[8] = {1, 2, 3, 4, 5, 6, 7, 8};
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int q = 0; q < 8; ++q) f[q] = __builtin_fma(f[q], a, b);
    }
    asm volatile("s_nop 0" :: "v"(f[0]), "v"(f[1]), "v"(f[2]), "v"(f[3]), "v"(f[4]), "v"(f[5]), "v"(f[6]), "v"(f[7]));
    long long t3 = __builtin_amdgcn_s_memtime();
    if (l == 0) { cyc[blockIdx.x * 3] = t1 - t0; cyc[blockIdx.x * 3 + 1] = t2 - t1; cyc[blockIdx.x * 3 + 2] = t3 - t2; }
    out[blockIdx.x * 64 + l] = c0[0] + c1[1] + c2[2] + c3[3] + d0[0] + f[0] + f[7];
}
int main() {
    double* out; long long* cyc;
    hipMalloc(&out, 1024 * 64 * 8); hipMalloc(&cyc, 1024 * 3 * 8);
    const int n = 2000;
    for (int blocks : {1, 1024}) {
        for (int w : {64, 256, 512}) {           // waves per workgroup 1 / 4 (one per SIMD) / 8 (two per SIMD)
            hipLaunchKernelGGL(k_rate, dim3(blocks), dim3(w), 0, 0, out, cyc, n);
            hipDeviceSynchronize();
            long long h[3]; hipMemcpy(h, cyc, 24, hipMemcpyDeviceToHost);
            printf("%4d workgroups x %d waves: mfma_f64_16x16x4 %.1f cycles each (4 accumulators), %.1f (one accumulator); v_fma_f64 %.2f cycles each\n",
                   blocks, w / 64, (double)h[0] / (4.0 * n), (double)h[1] / (4.0 * n), (double)h[2] / (8.0 * n));
        }
    }
    printf("per 64-walker tile at D = 32: VALU symmetric form 560 FMAs per lane of ONE wave-equivalent; matrix form 64 MFMAs (+ row dot)\n");
    return 0;
}
