// Issue cost of Philox4x32-10 written with v_mul_hi_u32 + v_mul_lo_u32 (two quarter-rate multiplies per product) against the
// 64-bit product form (one v_mad_u64_u32 per product).   hipcc --offload-arch=gfx950 -O3 philox_rate.hip -o philox_rate
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
struct u4 { uint32_t x, y, z, w; };
template <int FORM>
__device__ __forceinline__ u4 philox(u4 c, uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint32_t hi0, lo0, hi1, lo1;
        if (FORM == 0) {
            hi0 = __umulhi(0xD2511F53u, c.x); lo0 = 0xD2511F53u * c.x;
            hi1 = __umulhi(0xCD9E8D57u, c.z); lo1 = 0xCD9E8D57u * c.z;
        } else {
            const uint64_t p0 = (uint64_t)0xD2511F53u * c.x, p1 = (uint64_t)0xCD9E8D57u * c.z;
            hi0 = (uint32_t)(p0 >> 32); lo0 = (uint32_t)p0; hi1 = (uint32_t)(p1 >> 32); lo1 = (uint32_t)p1;
        }
        c = u4{hi1 ^ c.y ^ k0, lo1, hi0 ^ c.w ^ k1, lo0};
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return c;
}
template <int FORM>
__global__ void k(uint32_t* out, uint32_t s, int n, long long* cyc) {
    u4 c{threadIdx.x, blockIdx.x, s, 7u};
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i) c = philox<FORM>(c, s + i, s ^ 5u);
    const long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = c.x ^ c.y ^ c.z ^ c.w;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
int main() {
    uint32_t* out; long long* cyc; long long h;
    hipMalloc(&out, 1 << 24); hipMalloc(&cyc, 8);
    for (int waves = 1; waves <= 4; waves *= 2) {
        for (int form = 0; form < 2; ++form) {
            for (int rep = 0; rep < 2; ++rep) {
                if (form == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(256 * waves), 0, 0, out, 3u, 100, cyc);
                else hipLaunchKernelGGL(k<1>, dim3(256), dim3(256 * waves), 0, 0, out, 3u, 100, cyc);
                hipDeviceSynchronize();
            }
            hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
            printf("form %d (%s), %d wave(s) per SIMD: %.0f cycles per Philox4x32-10 call\n", form, form ? "v_mad_u64_u32" : "mul_hi + mul_lo",
                   waves, h / 100.0);
        }
    }
    return 0;
}
