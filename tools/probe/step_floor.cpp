// Probe (round 4): the fixed cost of a short hens_step call measured from C through the C ABI - no Python, no torch in the process.
//   hipcc -O2 tools/probe/step_floor.cpp -Iinclude -Leryn_amd/lib -lhipensemble -Wl,-rpath,$PWD/eryn_amd/lib -o build_ab/step_floor
// Config 2 (16 x 4096 x 32, dense Gaussian), blocks of K iterations between synchronisations, for K = 1, 2, 5, 10, 20, 40, 200:
// a straight line through (K, wall time) gives the fixed cost (intercept) and the steady rate (slope).
#include "hipensemble.h"
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
#include <algorithm>
#include <cmath>

using clk = std::chrono::high_resolution_clock;
static double us_since(clk::time_point t0) { return std::chrono::duration<double, std::micro>(clk::now() - t0).count(); }
#define CH(x) do { int r_ = (x); if (r_) { printf("%s -> %d: %s\n", #x, r_, hens_last_error(ctx)); return 1; } } while (0)

int main(int argc, char** argv) {
    const int T = 16, W = 4096, D = 32;
    const bool devsync = argc > 1 && atoi(argv[1]) == 1;      // also hipDeviceSynchronize after hens_synchronize (bench.py's torch sync)
    hens_config cfg{};
    cfg.ntemps = T; cfg.nwalkers = W; cfg.ndim = D; cfg.rung_begin = 0; cfg.rung_end = T; cfg.device_id = 0;
    cfg.likelihood_kind = HENS_LIKE_GAUSS_DENSE; cfg.tempered = 1; cfg.adaptive = 1; cfg.stop_adaptation = -1;
    cfg.a = 2.0; cfg.fill_value = -1e300; cfg.adaptation_lag = 10000; cfg.adaptation_time = 100; cfg.seed = 2024;
    hens_ctx* ctx = nullptr;
    CH(hens_create(&cfg, &ctx));
    std::mt19937_64 g(1);
    std::normal_distribution<double> nd;
    std::vector<double> lo(D, -50.0), hi(D, 50.0), mu(D), prec((size_t)D * D, 0.0), x((size_t)T * W * D), betas(T);
    for (int i = 0; i < D; ++i) { mu[i] = 0.1 * nd(g); prec[(size_t)i * D + i] = 1.0 + 0.1 * i; }
    for (int i = 0; i < D; ++i) for (int j = 0; j < i; ++j) prec[(size_t)i * D + j] = prec[(size_t)j * D + i] = 0.01 * nd(g);
    for (double& v : x) v = nd(g);
    for (int t = 0; t < T; ++t) betas[t] = std::pow(1.3, -t);
    CH(hens_set_prior_box(ctx, lo.data(), hi.data(), D * std::log(1.0 / 100.0)));
    CH(hens_set_gaussian(ctx, mu.data(), prec.data()));
    CH(hens_upload_state(ctx, x.data(), nullptr, nullptr, betas.data()));
    CH(hens_eval_state(ctx));
    CH(hens_step(ctx, 500));
    CH(hens_synchronize(ctx));
    std::vector<double> ks, ts;
    for (int K : {1, 2, 5, 10, 20, 40, 200, 2000}) {
        std::vector<double> v, vh;
        const int reps = K >= 200 ? 10 : 60;
        for (int r = 0; r < reps; ++r) {
            (void)hipDeviceSynchronize();
            auto t0 = clk::now();
            CH(hens_step(ctx, K));
            vh.push_back(us_since(t0));
            CH(hens_synchronize(ctx));
            if (devsync) (void)hipDeviceSynchronize();
            v.push_back(us_since(t0));
        }
        std::sort(v.begin(), v.end());
        const double med = v[v.size() / 2];
        std::sort(vh.begin(), vh.end());
        printf("K = %4d: median %8.1f us per call (min %8.1f) = %6.2f us per iteration   [host time inside hens_step %7.1f us]\n", K, med, v[0], med / K, vh[vh.size() / 2]);
        if (K <= 40) { ks.push_back(K); ts.push_back(med); }
    }
    double sx = 0, sy = 0, sxx = 0, sxy = 0; const double n = (double)ks.size();
    for (size_t i = 0; i < ks.size(); ++i) { sx += ks[i]; sy += ts[i]; sxx += ks[i] * ks[i]; sxy += ks[i] * ts[i]; }
    const double slope = (n * sxy - sx * sy) / (n * sxx - sx * sx), icpt = (sy - slope * sx) / n;
    printf("fit over K <= 40: %.2f us per iteration + %.1f us per call%s\n", slope, icpt, devsync ? "  (hens_synchronize + hipDeviceSynchronize)" : "");
    hens_destroy(ctx);
    return 0;
}
