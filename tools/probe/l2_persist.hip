// Does an XCD's L2 keep its lines from one kernel launch to the next, and what does a write from another XCD do to them?
//   hipcc --offload-arch=gfx950 -O3 tools/probe/l2_persist.hip -o /tmp/l2_persist && /tmp/l2_persist
// Workgroup L (dispatched round-robin to the 8 XCDs: XCD = L mod 8) reads chunk (L + rot) of a buffer and sums it.
//   A  rot = 0 every launch        : every chunk is re-read by the XCD that read it last time
//   B  rot = launch index          : every chunk is re-read by ANOTHER XCD
//   C  rot = 0, but between two reads a kernel whose workgroups are shifted by one XCD rewrites every chunk (sc1 stores):
//      the readers must see the new values; the time says whether their lines survived
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
constexpr int CHUNK = 16384;   // bytes per workgroup
typedef double dv2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void k_read(const double* buf, double* out, int nchunks, int rot, int passes = 1) {
    const int c = (blockIdx.x + rot) % nchunks;
    const dv2* p = reinterpret_cast<const dv2*>(buf + (size_t)c * (CHUNK / 8));
    double s = 0.0;
    for (int q = 0; q < passes; ++q)                 // (passes > 1: the later passes hit the XCD's L2 - or its L1: 16 KB per workgroup)
        for (int i = threadIdx.x; i < CHUNK / 16; i += 256) { const dv2 v = __builtin_nontemporal_load(p + i); s += v[0] + v[1] + q; }
    for (int m = 32; m; m >>= 1) s += __shfl_xor(s, m);
    if ((threadIdx.x & 63) == 0) atomicAdd(&out[blockIdx.x], s);
}
__global__ __launch_bounds__(256) void k_write(double* buf, int nchunks, int rot, double val) {
    const int c = (blockIdx.x + rot) % nchunks;
    double* p = buf + (size_t)c * (CHUNK / 8);
    for (int i = threadIdx.x; i < CHUNK / 8; i += 256) __hip_atomic_store(p + i, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
int main() {
    for (int mb : {8, 16, 24, 64}) {
        const int nchunks = mb * 1024 * 1024 / CHUNK;
        double *buf, *out;
        CK(hipMalloc(&buf, (size_t)nchunks * CHUNK)); CK(hipMalloc(&out, nchunks * 8));
        CK(hipMemset(buf, 0, (size_t)nchunks * CHUNK));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int mode = 0; mode < 3; ++mode) {
            const int reps = 40;
            for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k_read, dim3(nchunks), dim3(256), 0, 0, buf, out, nchunks, 0);
            CK(hipMemset(out, 0, nchunks * 8));
            CK(hipDeviceSynchronize());
            float total = 0;
            for (int i = 0; i < reps; ++i) {
                if (mode == 2) hipLaunchKernelGGL(k_write, dim3(nchunks), dim3(256), 0, 0, buf, nchunks, 1, (double)(i + 1));
                CK(hipEventRecord(e0));
                hipLaunchKernelGGL(k_read, dim3(nchunks), dim3(256), 0, 0, buf, out, nchunks, mode == 1 ? i + 1 : 0);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); total += ms;
            }
            std::vector<double> h(nchunks);
            CK(hipMemcpy(h.data(), out, nchunks * 8, hipMemcpyDeviceToHost));
            bool ok = true;
            if (mode == 2) { const double want = (CHUNK / 8) * (double)reps * (reps + 1) / 2; for (double v : h) ok = ok && v == want; }
            printf("%3d MB  mode %c: %7.2f us per read launch = %6.2f TB/s%s\n", mb, "ABC"[mode], total / reps * 1e3,
                   (double)nchunks * CHUNK / (total / reps * 1e-3) / 1e12, mode == 2 ? (ok ? "   (every reader saw the new values)" : "   STALE VALUES READ") : "");
        }
        {   // D: one launch reading every chunk 1 and 9 times: (t9 - t1) / 8 = a pass out of the cache hierarchy
            float t[2];
            for (int k = 0; k < 2; ++k) {
                CK(hipEventRecord(e0));
                for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k_read, dim3(nchunks), dim3(256), 0, 0, buf, out, nchunks, 0, k ? 9 : 1);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                CK(hipEventElapsedTime(&t[k], e0, e1));
            }
            printf("%3d MB  mode D: a further pass inside the launch %7.2f us = %6.2f TB/s\n", mb, (t[1] - t[0]) / 10 / 8 * 1e3,
                   (double)nchunks * CHUNK / ((t[1] - t[0]) / 10 / 8 * 1e-3) / 1e12);
        }
        CK(hipFree(buf)); CK(hipFree(out));
    }
    return 0;
}
