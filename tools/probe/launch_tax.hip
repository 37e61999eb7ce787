// Probe: what one dependent launch costs on this box, by footprint and launch flavour (round 3).
//   hipcc --offload-arch=gfx950 -O3 tools/probe/launch_tax.hip -o build_ab/launch_tax && build_ab/launch_tax
// A chain of kernels that each spin for a known time on the 100 MHz wall clock; tax = time per launch - spin.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <chrono>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

struct Big { void* p[40]; double d[8]; int i[16]; };   // ~450 bytes of kernel arguments, like StretchArgs
struct Small { int i[4]; };

template <int NT, bool FAT, class ARG>
__global__ __launch_bounds__(NT) void k_spin(const ARG a, long long ticks) {
    extern __shared__ char smem[];
    if (FAT) asm volatile("v_mov_b32 v100, 0" ::: "v100");          // forces a 101+-VGPR allocation
    if (ticks > 0) {
        const long long t0 = wall_clock64();
        while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(4);
    }
    if (a.i[0] == 12345) smem[threadIdx.x] = 1;
}

// the same spin, then the stores a stepping kernel ends with: kind 1 = plain 16-B stores (dirty L2 lines), 2 = sc1 write-through,
// 3 = plain + one atomicAdd per lane of wave 0; `per_wg` 16-byte chunks per workgroup at scattered 256-B rows
typedef double dv2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(512) void k_spin_store(const Big a, long long ticks, double* out, unsigned* cnt, int kind, int per_wg) {
    extern __shared__ char smem[];
    asm volatile("v_mov_b32 v100, 0" ::: "v100");
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(4);
    const int tid = threadIdx.x;
    if (tid < per_wg) {
        const unsigned row = (blockIdx.x * 2654435761u + (tid >> 4) * 40503u) & 65535u;       // scattered rows of 32 doubles
        double* p = out + (size_t)row * 32 + (tid & 15) * 2;
        const dv2 v = {(double)tid, 1.0};
        if (kind == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
        else *reinterpret_cast<dv2*>(p) = v;
    }
    if (kind == 3 && tid < 64) atomicAdd(&cnt[(blockIdx.x * 64 + tid) & 65535], 1u);
    if (a.i[0] == 12345) smem[threadIdx.x] = 1;
}

template <class F>
double time_chain(hipStream_t s, int n, F launch) {
    for (int i = 0; i < 100; ++i) launch();
    hipStreamSynchronize(s);
    auto t0 = std::chrono::high_resolution_clock::now();
    for (int i = 0; i < n; ++i) launch();
    hipStreamSynchronize(s);
    auto t1 = std::chrono::high_resolution_clock::now();
    return std::chrono::duration<double, std::micro>(t1 - t0).count() / n;
}

template <int NT, bool FAT, class ARG>
int run(hipStream_t s, const char* name, int wgs, int lds, ARG a) {
    for (long long us : {0LL, 6LL}) {
        const long long ticks = us * 100;
        double te = time_chain(s, 3000, [&] { hipLaunchKernelGGL((k_spin<NT, FAT, ARG>), dim3(wgs), dim3(NT), lds, s, a, ticks); });
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
        for (int i = 0; i < 40; ++i) hipLaunchKernelGGL((k_spin<NT, FAT, ARG>), dim3(wgs), dim3(NT), lds, s, a, ticks);
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        double tg = time_chain(s, 100, [&] { hipGraphLaunch(ge, s); }) / 40;
        printf("  %-44s %4d WGs x %4d thr, lds %5d, spin %lld us:  eager %.2f (tax %.2f)   graph %.2f (tax %.2f)\n", name, wgs, NT, lds, us,
               te, te - us, tg, tg - us);
        hipGraphExecDestroy(ge); hipGraphDestroy(g);
    }
    return 0;
}

int main() {
    hipStream_t s, sp;
    CK(hipStreamCreate(&s));
    int lo, hi;
    CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    CK(hipStreamCreateWithPriority(&sp, hipStreamNonBlocking, hi));
    Big a{}; Small b{};
    printf("chain of dependent launches on one stream, us per launch (host wall / n)\n");
    run<512, false, Small>(s, "thin (few VGPRs), small args", 512, 0, b);
    run<512, false, Big>(s, "thin, 450-B args", 512, 0, a);
    run<512, false, Big>(s, "thin, 450-B args, 36 KB LDS", 512, 36000, a);
    run<512, true, Big>(s, "fat (101 VGPRs), 450-B args, 36 KB LDS", 512, 36000, a);
    run<512, true, Big>(sp, "same on a high-priority non-blocking stream", 512, 36000, a);
    run<256, true, Big>(s, "fat, 256 thr x 1024 WGs, 18 KB LDS", 1024, 18000, a);
    run<1024, true, Big>(s, "fat, 1024 thr x 256 WGs, 72 KB LDS", 256, 72000, a);
    run<512, true, Big>(s, "fat, 256 WGs", 256, 36000, a);
    double* out; unsigned* cnt;
    CK(hipMalloc(&out, (size_t)65536 * 256)); CK(hipMalloc(&cnt, 65536 * 4));
    CK(hipMemset(cnt, 0, 65536 * 4));
    for (int kind : {1, 2, 3})
        for (int per : {64, 256, 512}) {
            double t = time_chain(s, 3000, [&] { hipLaunchKernelGGL(k_spin_store, dim3(512), dim3(512), 36000, s, a, 600LL, out, cnt, kind, per); });
            printf("  spin 6 us + %s, %3d x 16 B per WG (%4.1f MB per launch): %.2f per launch (tax %.2f)\n",
                   kind == 1 ? "plain stores      " : (kind == 2 ? "sc1 stores        " : "plain + 64 atomics"), per, per * 16 * 512 / 1e6, t, t - 6.0);
        }
    return 0;
}
