// hens_kernels.h - gfx950 (CDNA4, wave64) device code of libhipensemble.
//
// Compiled with -ffp-contract=off: every a*b+c below rounds twice like NumPy
// unless it is written as an explicit fma().  That is what makes the proposal
// q = c - (c - s) * zz (stretch.py:143-145) and the accept arithmetic
// (red_blue.py:292) bit-identical to the reference.
//
// Data layout in HBM (one context = the ladder shard [rung_begin, rung_end)):
//   pool   f64 [2 * Tl * W][D]   walker rows, AoS (a row is one contiguous 8*D-byte burst).
//                                Every walker (tl, w) owns two "home" rows, tl*W+w and
//                                Tl*W + tl*W+w; iteration parity p writes proposals' results
//                                into home_p and reads through `loc`, so a stretch step is
//                                read-own + read-complement + write-own with no in-place hazard
//                                and the PT cascade never moves a row: it permutes `loc`.
//   loc    i32 [Tl * W]          pool row currently holding walker (tl, w)
//   L, P   f64 [Tl * W]          log-likelihood / log-prior (double buffered for the cascade)
//   betas  f64 [T]               full ladder
#pragma once
#include <hip/hip_runtime.h>
#ifndef HENS_ABLATE
#define HENS_ABLATE 0   // timing experiments only (tools/ablate.sh): bit0 no quad form, bit1 no logs, bit2 no row write, bit3 no philox
#endif
#include <stdint.h>

namespace hens {

constexpr int TILE = 64;                 // walkers per workgroup = one wavefront of walker-lanes
constexpr unsigned FLAG_NONFINITE_X = 1u;   // inf/NaN coordinate seen (ensemble.py:1258-1262)
constexpr unsigned FLAG_NAN_LOGL = 2u;      // NaN likelihood (red_blue.py:279-281)

enum { LIKE_DENSE = 0, LIKE_DIAG = 1, LIKE_ROSEN = 2 };
enum { MODE_PARITY = 0, MODE_PHILOX = 1, MODE_EVAL = 2 };

// ---------------------------------------------------------------------------------------------
// Philox4x32-10 (Salmon et al. 2011), counter-based: draws are a pure function of
// (seed, iteration, purpose, walker), so any kernel / any rank regenerates them identically.
// ---------------------------------------------------------------------------------------------
struct u4 { uint32_t x, y, z, w; };

__device__ __forceinline__ u4 philox4x32_10(u4 c, uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c.x), lo0 = 0xD2511F53u * c.x;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c.z), lo1 = 0xCD9E8D57u * c.z;
        c = u4{hi1 ^ c.y ^ k0, lo1, hi0 ^ c.w ^ k1, lo0};
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return c;
}
__device__ __forceinline__ double u01(uint32_t hi, uint32_t lo) {   // 53-bit uniform in [0, 1)
    const uint64_t v = ((uint64_t)hi << 32) | lo;
    return (double)(v >> 11) * (1.0 / 9007199254740992.0);
}
enum : uint32_t { PURPOSE_STRETCH0 = 0, PURPOSE_STRETCH1 = 1, PURPOSE_STRETCH_ACC = 2,
                  PURPOSE_SPLIT = 8, PURPOSE_PTPERM = 9, PURPOSE_PTU = 10 };

// ---------------------------------------------------------------------------------------------
// Stretch half-step: propose + box prior + likelihood + tempered MH test + update, fused.
// ---------------------------------------------------------------------------------------------
struct StretchArgs {
    double* pool;
    int32_t* loc;
    double* L;
    double* P;
    const double* betas;       // [T] or nullptr when not tempered
    const int32_t* order;      // [Tl][W]: first N0 entries = walkers of split 0, rest = split 1
    uint32_t* accepted;        // [Tl][W] cumulative accept counts
    uint8_t* keep_out;         // [Tl][Ns] or nullptr
    const int64_t* rint;       // parity draws [Tl][Ns]
    const double* u_zz;
    const double* u_acc;
    const double* lo;
    const double* hi;
    const double* mu;
    const double* prec;
    const uint64_t* clock;     // device iteration counter
    unsigned* flags;
    unsigned long long* trace;  // debug: per-workgroup phase timestamps (s_memtime), or nullptr
    double a, logp_in, fill, rosen_a, rosen_b;
    uint64_t seed;
    int32_t Tl, W, D, split, N0, rung_begin, home_off, tempered, RS;
};

template <int DT> struct DimOf { static __device__ __forceinline__ int get(int d) { return DT; } };
template <> struct DimOf<0> { static __device__ __forceinline__ int get(int d) { return d; } };

// One workgroup = TILE (64) walkers of one rung, NW wavefronts.
//   phase A  lane-per-walker : indices, draws, stretch factor            (wave 0)
//   phase B  lanes-over-d    : coalesced row gathers, q = c-(c-s)zz, box test by ballot -> LDS
//   phase C  lane-per-walker : quadratic form, rows of the precision matrix split over the NW
//                              waves and fed from SGPRs (scalar loads), q held in VGPRs
//   phase D  lane-per-walker : tempered MH test, L/P/loc/accept counters
//   phase E  lanes-over-d    : coalesced write of the new row (q if kept, old row otherwise)
template <int DT, int LIKE, int MODE, int NW>
__global__ __launch_bounds__(NW * 64) void k_stretch(const StretchArgs A) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int D = DimOf<DT>::get(A.D);
    const int RS = (DT > 0) ? ((DT % 2 == 0) ? DT + 2 : DT) : A.RS;   // LDS row stride (doubles)
    constexpr int NT = NW * 64;
    double* qtile = reinterpret_cast<double*>(smem_raw);                 // [TILE][RS]
    double* s_zz = qtile + TILE * RS;                                    // [TILE]
    double* s_part = s_zz + TILE;                                        // [NW][TILE]
    int32_t* s_rs = reinterpret_cast<int32_t*>(s_part + NW * TILE);      // [TILE] own row
    int32_t* s_rc = s_rs + TILE;                                         // [TILE] complement row
    int32_t* s_dst = s_rc + TILE;                                        // [TILE] destination row
    int32_t* s_flag = s_dst + TILE;                                      // [TILE] bit0 inbox, bit1 keep, bit2 valid

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tl = blockIdx.y;
    const int W = A.W;
    const int Ns = (MODE == MODE_EVAL) ? W : (A.split == 0 ? A.N0 : W - A.N0);
    const int Nc = W - Ns;
    const int s_off = (MODE == MODE_EVAL) ? 0 : (A.split == 0 ? 0 : A.N0);
    const int c_off = (A.split == 0 ? A.N0 : 0);
    const int k0 = blockIdx.x * TILE;

    // ---- phase A ---------------------------------------------------------------------------
    double factors = 0.0, lu = 0.0, Lold = 0.0, Pold = 0.0;
    int own = 0;
    bool valid = false;
    if (wv == 0) {
        const int k = k0 + lane;
        valid = k < Ns;
        double zz = 1.0;
        int rs = 0, rc = 0;
        if (valid) {
            if (MODE == MODE_EVAL) {
                own = k;
                rs = A.loc[tl * W + own];
                rc = rs;
            } else {
                own = A.order[tl * W + s_off + k];
                double uz, ua;
                int r;
                if (MODE == MODE_PARITY) {
                    r = (int)A.rint[(size_t)tl * Ns + k];
                    uz = A.u_zz[(size_t)tl * Ns + k];
                    ua = A.u_acc[(size_t)tl * Ns + k];
                } else {
                    const uint64_t it = A.clock[0];
                    const u4 ctr{(uint32_t)it, (uint32_t)(it >> 32),
                                 (uint32_t)((A.rung_begin + tl) * W + own), PURPOSE_STRETCH0};
                    const u4 d = philox4x32_10(ctr, (uint32_t)A.seed, (uint32_t)(A.seed >> 32));
                    r = (int)__umulhi(d.x, (uint32_t)Nc);
                    uz = u01(d.y, d.z);
                    u4 ctr2 = ctr;
                    ctr2.w = PURPOSE_STRETCH_ACC;
                    const u4 e = philox4x32_10(ctr2, (uint32_t)A.seed, (uint32_t)(A.seed >> 32));
                    ua = u01(e.x, e.y);
                }
                const int cw = A.order[tl * W + c_off + r];
                rs = A.loc[tl * W + own];
                rc = A.loc[tl * W + cw];
                zz = (A.a - 1.0) * uz + 1.0;          // stretch.py:129-132
                zz = zz * zz / A.a;
                factors = ((double)D - 1.0) * log(zz);   // stretch.py:223
                lu = log(ua);                            // red_blue.py:294
                Lold = A.L[tl * W + own];
                Pold = A.P[tl * W + own];
            }
        }
        s_zz[lane] = zz;
        s_rs[lane] = rs;
        s_rc[lane] = rc;
        s_dst[lane] = A.home_off + tl * W + own;
        s_flag[lane] = valid ? 4 : 0;
    }
    __syncthreads();

    // ---- phase B: lanes over d -------------------------------------------------------------
    // VEC doubles per lane; LPR lanes per row (power of two <= 64); RPP rows per pass.
    const int VEC = (D % 2 == 0) ? 2 : 1;
    const int chunks = D / VEC;
    int LPR = 1;
    while (LPR < chunks && LPR < 64) LPR <<= 1;
    const int RPP = NT / LPR;
    const int jl = tid & (LPR - 1);
    const int rsub = tid / LPR;
    const double* __restrict__ pool_r = A.pool;
    for (int r0 = 0; r0 < TILE; r0 += RPP) {
        const int r = r0 + rsub;
        const bool rvalid = (r < TILE) && (s_flag[r] & 4) != 0;   // RPP may exceed TILE for tiny D
        bool ok = true, finite = true;
        if (rvalid) {
            const double zz = s_zz[r];
            const double* ps = pool_r + (size_t)s_rs[r] * D;
            const double* pc = pool_r + (size_t)s_rc[r] * D;
            for (int ch = jl; ch < chunks; ch += LPR) {
                const int e = ch * VEC;
                if (VEC == 2) {
                    const double2 sv = *reinterpret_cast<const double2*>(ps + e);
                    double2 qv;
                    if (MODE == MODE_EVAL) {
                        qv = sv;
                    } else {
                        const double2 cv = *reinterpret_cast<const double2*>(pc + e);
                        qv.x = cv.x - (cv.x - sv.x) * zz;     // stretch.py:143,145
                        qv.y = cv.y - (cv.y - sv.y) * zz;
                    }
                    const double2 lov = *reinterpret_cast<const double2*>(A.lo + e);
                    const double2 hiv = *reinterpret_cast<const double2*>(A.hi + e);
                    ok = ok && (qv.x >= lov.x) && (qv.x <= hiv.x) && (qv.y >= lov.y) && (qv.y <= hiv.y);
                    finite = finite && (fabs(qv.x) < INFINITY) && (fabs(qv.y) < INFINITY);
                    *reinterpret_cast<double2*>(qtile + r * RS + e) = qv;
                } else {
                    const double sv = ps[e];
                    double qv;
                    if (MODE == MODE_EVAL) {
                        qv = sv;
                    } else {
                        const double cv = pc[e];
                        qv = cv - (cv - sv) * zz;
                    }
                    ok = ok && (qv >= A.lo[e]) && (qv <= A.hi[e]);
                    finite = finite && (fabs(qv) < INFINITY);
                    qtile[r * RS + e] = qv;
                }
            }
        }
        // row-wide AND over the LPR lanes of a row via wavefront ballot (prior.py:80-88)
        const unsigned long long bad = __ballot(!ok);
        const unsigned long long nonfin = __ballot(!finite);
        const int gshift = lane & ~(LPR - 1);
        const unsigned long long gmask = (LPR == 64) ? ~0ull : (((1ull << LPR) - 1ull) << gshift);
        if (jl == 0 && rvalid) {
            if ((bad & gmask) == 0ull) atomicOr(&s_flag[r], 1);
            if ((nonfin & gmask) != 0ull) atomicOr(A.flags, FLAG_NONFINITE_X);
        }
    }
    __syncthreads();

    // ---- phase C: likelihood, lane per walker, precision rows split over waves ---------------
    {
        const bool inbox = (s_flag[lane] & 1) != 0;
        double part = 0.0;
        // model constants are read-only for the whole launch: address them through the constant
        // address space so wave-uniform indices become SGPR scalar loads (s_load_dwordx*)
        typedef const __attribute__((address_space(4))) double* cptr_t;
        const cptr_t mu = (cptr_t)(uintptr_t)A.mu;
        const cptr_t prec = (cptr_t)(uintptr_t)A.prec;
        const double* qrow = qtile + lane * RS;
        if (LIKE == LIKE_ROSEN) {
            if (wv == 0 && inbox) {
                double acc = 0.0;
                for (int i = 0; i + 1 < D; ++i) {
                    const double x0 = qrow[i], x1 = qrow[i + 1];
                    const double t1 = x1 - x0 * x0, t2 = A.rosen_a - x0;
                    acc += A.rosen_b * (t1 * t1) + t2 * t2;
                }
                part = 2.0 * acc;      // phase D multiplies by -0.5
            }
        } else if (DT > 0) {
            constexpr int DC = (DT > 0) ? DT : 1;
            constexpr int RB = (DC + NW - 1) / NW;
            if (inbox) {
                double qreg[DC];
#pragma unroll
                for (int k = 0; k < DC; ++k) qreg[k] = qrow[k] - mu[k];
                const int i0 = wv * RB;
                if (LIKE == LIKE_DENSE) {
#pragma unroll 2
                    for (int ii = 0; ii < RB; ++ii) {
                        const int i = i0 + ii;
                        if (i < DC) {
                            const cptr_t prow = prec + (size_t)i * DC;
                            double y0 = 0.0, y1 = 0.0;
#pragma unroll
                            for (int k = 0; k + 1 < DC; k += 2) {
                                y0 = fma(prow[k], qreg[k], y0);
                                y1 = fma(prow[k + 1], qreg[k + 1], y1);
                            }
                            if (DC & 1) y0 = fma(prow[DC - 1], qreg[DC - 1], y0);
                            part = fma(qrow[i] - mu[i], y0 + y1, part);
                        }
                    }
                } else {
                    for (int ii = 0; ii < RB; ++ii) {
                        const int i = i0 + ii;
                        if (i < DC) {
                            const double di = qrow[i] - mu[i];
                            part = fma(di * prec[i], di, part);
                        }
                    }
                }
            }
        } else {
            const int RB = (D + NW - 1) / NW;
            if (inbox) {
                const int i0 = wv * RB;
                for (int ii = 0; ii < RB; ++ii) {
                    const int i = i0 + ii;
                    if (i < D) {
                        const double di = qrow[i] - mu[i];
                        if (LIKE == LIKE_DENSE) {
                            double y = 0.0;
                            for (int k = 0; k < D; ++k) y = fma(prec[(size_t)i * D + k], qrow[k] - mu[k], y);
                            part = fma(di, y, part);
                        } else {
                            part = fma(di * prec[i], di, part);
                        }
                    }
                }
            }
        }
        s_part[wv * TILE + lane] = part;
    }
    __syncthreads();

    // ---- phase D: accept / update, lane per walker -------------------------------------------
    if (wv == 0 && valid) {
        const bool inbox = (s_flag[lane] & 1) != 0;
        double acc = 0.0;
#pragma unroll
        for (int w2 = 0; w2 < NW; ++w2) acc += s_part[w2 * TILE + lane];
        double logl = inbox ? -0.5 * acc : A.fill;             // ensemble.py:1486-1513
        if (logl != logl) {                                    // red_blue.py:279-281
            logl = -1e300;
            atomicOr(A.flags, FLAG_NAN_LOGL);
        }
        const double logp = inbox ? A.logp_in : -INFINITY;     // prior.py:80-88
        const size_t gi = (size_t)tl * W + own;
        if (MODE == MODE_EVAL) {
            A.L[gi] = logl;
            A.P[gi] = logp;
        } else {
            double logP, prevP;
            if (A.tempered) {                                  // tempering.py:304-306,343-349
                const double beta = A.betas[A.rung_begin + tl];
                double lt = logl * beta;
                if (lt != lt) lt = -INFINITY;
                logP = lt + logp;
                double lo_ = Lold * beta;
                if (lo_ != lo_) lo_ = -INFINITY;
                prevP = lo_ + Pold;
            } else {                                           // move.py:443-457
                logP = logl + logp;
                prevP = Lold + Pold;
            }
            const double lnpdiff = factors + logP - prevP;     // red_blue.py:292
            const bool keep = lnpdiff > lu;                    // red_blue.py:294
            if (keep) {                                        // move.py:513-532
                A.L[gi] = logl;
                A.P[gi] = (fabs(logp) == INFINITY) ? 0.0 : logp;
                atomicAdd(&A.accepted[gi], 1u);
                atomicOr(&s_flag[lane], 2);
            }
            A.loc[gi] = s_dst[lane];
            if (A.keep_out) A.keep_out[(size_t)tl * Ns + k0 + lane] = keep ? 1 : 0;
        }
    }
    if (MODE == MODE_EVAL) return;
    __syncthreads();

    // ---- phase E: write rows, lanes over d ----------------------------------------------------
    double* __restrict__ pool_w = A.pool;
    for (int r0 = 0; r0 < TILE; r0 += RPP) {
        const int r = r0 + rsub;
        if (r >= TILE) continue;
        const int fl = s_flag[r];
        if (!(fl & 4)) continue;
        const bool keep = (fl & 2) != 0;
        double* pd = pool_w + (size_t)s_dst[r] * D;
        const double* ps = pool_r + (size_t)s_rs[r] * D;
        for (int ch = jl; ch < chunks; ch += LPR) {
            const int e = ch * VEC;
            if (VEC == 2) {
                const double2 v = keep ? *reinterpret_cast<const double2*>(qtile + r * RS + e)
                                       : *reinterpret_cast<const double2*>(ps + e);
                *reinterpret_cast<double2*>(pd + e) = v;
            } else {
                pd[e] = keep ? qtile[r * RS + e] : ps[e];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Fast path for power-of-two row widths (D = 8, 16, 32, 64): same five phases, but
//   * every row chunk a thread will touch is loaded up front (NPASS x 2 x 16 B per thread in
//     flight) instead of one pass at a time -> the kernel is latency-bound at config-2 size, so
//     memory-level parallelism is what buys time;
//   * the old row stays in registers for the write-back (no re-read on reject);
//   * the two log() calls of the accept test are issued while those loads are in flight.
// ---------------------------------------------------------------------------------------------
template <int DT, int LIKE, int MODE, int NW, bool PLDS>
__global__ __launch_bounds__(NW * 64) void k_stretch_fast(const StretchArgs A) {
    static_assert(DT == 8 || DT == 16 || DT == 32 || DT == 64 || DT == 128, "power-of-two row width");
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    constexpr int D = DT;
    constexpr int RS = DT + 2;                 // LDS row stride (doubles): conflict-free b128 reads
    constexpr int NT = NW * 64;
    constexpr int LPR = DT / 2;                // lanes per row, 16 B each
    static_assert(LPR <= 64, "row wider than a wavefront");
    constexpr int RPP = NT / LPR;              // rows per pass
    constexpr int NPASS = (TILE + RPP - 1) / RPP;
    double* qtile = reinterpret_cast<double*>(smem_raw);                 // [TILE][RS]
    double* s_zz = qtile + TILE * RS;                                    // [TILE]
    double* s_part = s_zz + TILE;                                        // [NW][TILE]
    int32_t* s_rs = reinterpret_cast<int32_t*>(s_part + NW * TILE);      // [TILE]
    int32_t* s_rc = s_rs + TILE;
    int32_t* s_dst = s_rc + TILE;
    int32_t* s_flag = s_dst + TILE;                                      // bit0 inbox, bit1 keep, bit2 valid
    double* s_prec = reinterpret_cast<double*>(s_flag + TILE);           // [DT][DT] when PLDS

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tl = blockIdx.y;
    const int W = A.W;
    const int Ns = (MODE == MODE_EVAL) ? W : (A.split == 0 ? A.N0 : W - A.N0);
    const int Nc = W - Ns;
    const int s_off = (MODE == MODE_EVAL) ? 0 : (A.split == 0 ? 0 : A.N0);
    const int c_off = (A.split == 0 ? A.N0 : 0);
    const int k0 = blockIdx.x * TILE;
#define HENS_TRACE(i) do { if (A.trace && tid == 0) A.trace[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
    HENS_TRACE(0);

    if (PLDS && LIKE == LIKE_DENSE) {
        // stage the precision matrix in LDS while phase A/B wait on their gathers: phase C then has
        // no global latency at all (a cold scalar cache costs ~1 us per pair of rows otherwise)
        for (int i = tid * 2; i < DT * DT; i += NT * 2)
            *reinterpret_cast<double2*>(s_prec + i) = *reinterpret_cast<const double2*>(A.prec + i);
    }
    // ---- phase A (wave 0): indices and draws; the logs wait until the row loads are in flight ---
    double zz_own = 1.0, ua = 1.0, Lold = 0.0, Pold = 0.0;
    int own = 0;
    bool valid = false;
    if (wv == 0) {
        const int k = k0 + lane;
        valid = k < Ns;
        int rs = 0, rc = 0;
        if (valid) {
            if (MODE == MODE_EVAL) {
                own = k;
                rs = A.loc[tl * W + own];
                rc = rs;
            } else {
                own = A.order[tl * W + s_off + k];
                double uz;
                int r;
                if (MODE == MODE_PARITY) {
                    r = (int)A.rint[(size_t)tl * Ns + k];
                    uz = A.u_zz[(size_t)tl * Ns + k];
                    ua = A.u_acc[(size_t)tl * Ns + k];
                } else {
                    const uint64_t it = A.clock[0];
                    const u4 ctr{(uint32_t)it, (uint32_t)(it >> 32),
                                 (uint32_t)((A.rung_begin + tl) * W + own), PURPOSE_STRETCH0};
                    u4 d, e;
                    if (HENS_ABLATE & 8) {
                        d = u4{ctr.z * 2654435761u + ctr.x, ctr.z * 40503u, ctr.x * 7919u + ctr.z, 0u};
                        e = u4{d.y, d.x, 0u, 0u};
                    } else {
                        d = philox4x32_10(ctr, (uint32_t)A.seed, (uint32_t)(A.seed >> 32));
                        u4 ctr2 = ctr;
                        ctr2.w = PURPOSE_STRETCH_ACC;
                        e = philox4x32_10(ctr2, (uint32_t)A.seed, (uint32_t)(A.seed >> 32));
                    }
                    r = (int)__umulhi(d.x, (uint32_t)Nc);
                    uz = u01(d.y, d.z);
                    ua = u01(e.x, e.y);
                }
                const int cw = A.order[tl * W + c_off + r];
                rs = A.loc[tl * W + own];
                rc = A.loc[tl * W + cw];
                Lold = A.L[tl * W + own];
                Pold = A.P[tl * W + own];
                zz_own = (A.a - 1.0) * uz + 1.0;      // stretch.py:129-132
                zz_own = zz_own * zz_own / A.a;
            }
        }
        s_zz[lane] = zz_own;
        s_rs[lane] = rs;
        s_rc[lane] = rc;
        s_dst[lane] = A.home_off + tl * W + own;
        s_flag[lane] = valid ? 4 : 0;
    }
    HENS_TRACE(1);
    __syncthreads();
    HENS_TRACE(2);

    // ---- phase B: lanes over d, all loads first -------------------------------------------------
    const int jl = tid & (LPR - 1);
    const int rsub = tid / LPR;
    const double* __restrict__ pool_r = A.pool;
    double2 sreg[NPASS], creg[NPASS];
    bool rv[NPASS];
#pragma unroll
    for (int p = 0; p < NPASS; ++p) {
        const int r = p * RPP + rsub;
        rv[p] = (r < TILE) && (s_flag[r < TILE ? r : 0] & 4) != 0;
        sreg[p] = double2{0.0, 0.0};
        creg[p] = double2{0.0, 0.0};
        if (rv[p]) {
            sreg[p] = *reinterpret_cast<const double2*>(pool_r + (size_t)s_rs[r] * D + jl * 2);
            if (MODE != MODE_EVAL) creg[p] = *reinterpret_cast<const double2*>(pool_r + (size_t)s_rc[r] * D + jl * 2);
        }
    }
    const double2 lov = *reinterpret_cast<const double2*>(A.lo + jl * 2);
    const double2 hiv = *reinterpret_cast<const double2*>(A.hi + jl * 2);
    double factors = 0.0, lu = 0.0;
    if (MODE != MODE_EVAL && wv == 0 && !(HENS_ABLATE & 2)) {
        factors = ((double)D - 1.0) * log(zz_own);       // stretch.py:223
        lu = log(ua);                                    // red_blue.py:294
    }
#pragma unroll
    for (int p = 0; p < NPASS; ++p) {
        const int r = p * RPP + rsub;
        bool ok = true, finite = true;
        if (rv[p]) {
            double2 qv;
            if (MODE == MODE_EVAL) {
                qv = sreg[p];
            } else {
                const double zz = s_zz[r];
                qv.x = creg[p].x - (creg[p].x - sreg[p].x) * zz;     // stretch.py:143,145
                qv.y = creg[p].y - (creg[p].y - sreg[p].y) * zz;
            }
            ok = (qv.x >= lov.x) && (qv.x <= hiv.x) && (qv.y >= lov.y) && (qv.y <= hiv.y);
            finite = (fabs(qv.x) < INFINITY) && (fabs(qv.y) < INFINITY);
            *reinterpret_cast<double2*>(qtile + r * RS + jl * 2) = qv;
        }
        const unsigned long long bad = __ballot(!ok);               // prior.py:80-88, row-wide AND
        const unsigned long long nonfin = __ballot(!finite);
        const int gshift = lane & ~(LPR - 1);
        const unsigned long long gmask = (LPR == 64) ? ~0ull : (((1ull << (LPR & 63)) - 1ull) << gshift);
        if (jl == 0 && rv[p]) {
            if ((bad & gmask) == 0ull) atomicOr(&s_flag[r], 1);
            if ((nonfin & gmask) != 0ull) atomicOr(A.flags, FLAG_NONFINITE_X);
        }
    }
    HENS_TRACE(3);
    __syncthreads();
    HENS_TRACE(4);

    // ---- phase C: likelihood, lane per walker, precision rows split over the waves ----------------
    {
        const bool inbox = (s_flag[lane] & 1) != 0;
        double part = 0.0;
        typedef const __attribute__((address_space(4))) double* cptr_t;
        const cptr_t mu = (cptr_t)(uintptr_t)A.mu;
        const cptr_t prec = (cptr_t)(uintptr_t)A.prec;
        const double* qrow = qtile + lane * RS;
        if (LIKE == LIKE_ROSEN) {
            if (wv == 0 && inbox) {
                double acc = 0.0;
                for (int i = 0; i + 1 < D; ++i) {
                    const double x0 = qrow[i], x1 = qrow[i + 1];
                    const double t1 = x1 - x0 * x0, t2 = A.rosen_a - x0;
                    acc += A.rosen_b * (t1 * t1) + t2 * t2;
                }
                part = 2.0 * acc;
            }
        } else if (inbox && !(HENS_ABLATE & 1)) {
            constexpr int RB = (DT + NW - 1) / NW;
            const int i0 = wv * RB;
            if (LIKE == LIKE_DENSE) {
                double qreg[DT];
#pragma unroll
                for (int k = 0; k < DT; k += 2) {
                    const double2 v = *reinterpret_cast<const double2*>(qrow + k);
                    qreg[k] = v.x - mu[k];
                    qreg[k + 1] = v.y - mu[k + 1];
                }
#pragma unroll 2
                for (int ii = 0; ii < RB; ++ii) {
                    const int i = i0 + ii;
                    if (i < DT) {
                        double y0 = 0.0, y1 = 0.0;
                        if (PLDS) {
                            const double* prow = s_prec + i * DT;       // wave-uniform address: LDS broadcast
#pragma unroll
                            for (int k = 0; k < DT; k += 2) {
                                const double2 pv = *reinterpret_cast<const double2*>(prow + k);
                                y0 = fma(pv.x, qreg[k], y0);
                                y1 = fma(pv.y, qreg[k + 1], y1);
                            }
                        } else {
                            const cptr_t prow = prec + (size_t)i * DT;  // SGPR operands via s_load
#pragma unroll
                            for (int k = 0; k < DT; k += 2) {
                                y0 = fma(prow[k], qreg[k], y0);
                                y1 = fma(prow[k + 1], qreg[k + 1], y1);
                            }
                        }
                        part = fma(qrow[i] - mu[i], y0 + y1, part);
                    }
                }
            } else {
                for (int ii = 0; ii < RB; ++ii) {
                    const int i = i0 + ii;
                    if (i < DT) {
                        const double di = qrow[i] - mu[i];
                        part = fma(di * prec[i], di, part);
                    }
                }
            }
        }
        s_part[wv * TILE + lane] = part;
    }
    HENS_TRACE(5);
    __syncthreads();

    // ---- phase D: accept / update (wave 0) -------------------------------------------------------
    if (wv == 0 && valid) {
        const bool inbox = (s_flag[lane] & 1) != 0;
        double acc = 0.0;
#pragma unroll
        for (int w2 = 0; w2 < NW; ++w2) acc += s_part[w2 * TILE + lane];
        double logl = inbox ? -0.5 * acc : A.fill;             // ensemble.py:1486-1513
        if (logl != logl) {                                    // red_blue.py:279-281
            logl = -1e300;
            atomicOr(A.flags, FLAG_NAN_LOGL);
        }
        const double logp = inbox ? A.logp_in : -INFINITY;     // prior.py:80-88
        const size_t gi = (size_t)tl * W + own;
        if (MODE == MODE_EVAL) {
            A.L[gi] = logl;
            A.P[gi] = logp;
        } else {
            double logP, prevP;
            if (A.tempered) {                                  // tempering.py:304-306,343-349
                const double beta = A.betas[A.rung_begin + tl];
                double lt = logl * beta;
                if (lt != lt) lt = -INFINITY;
                logP = lt + logp;
                double lo_ = Lold * beta;
                if (lo_ != lo_) lo_ = -INFINITY;
                prevP = lo_ + Pold;
            } else {                                           // move.py:443-457
                logP = logl + logp;
                prevP = Lold + Pold;
            }
            const double lnpdiff = factors + logP - prevP;     // red_blue.py:292
            const bool keep = lnpdiff > lu;                    // red_blue.py:294
            if (keep) {                                        // move.py:513-532
                A.L[gi] = logl;
                A.P[gi] = (fabs(logp) == INFINITY) ? 0.0 : logp;
                atomicAdd(&A.accepted[gi], 1u);
                atomicOr(&s_flag[lane], 2);
            }
            A.loc[gi] = s_dst[lane];
            if (A.keep_out) A.keep_out[(size_t)tl * Ns + k0 + lane] = keep ? 1 : 0;
        }
    }
    if (MODE == MODE_EVAL) return;
    HENS_TRACE(6);
    __syncthreads();

    // ---- phase E: write rows (q if kept, the register copy of the old row otherwise) ---------------
    double* __restrict__ pool_w = A.pool;
#pragma unroll
    for (int p = 0; p < NPASS; ++p) {
        const int r = p * RPP + rsub;
        if (!rv[p] || (HENS_ABLATE & 4)) continue;
        const bool keep = (s_flag[r] & 2) != 0;
        const double2 v = keep ? *reinterpret_cast<const double2*>(qtile + r * RS + jl * 2) : sreg[p];
        *reinterpret_cast<double2*>(pool_w + (size_t)s_dst[r] * D + jl * 2) = v;
    }
    HENS_TRACE(7);
#undef HENS_TRACE
}

// ---------------------------------------------------------------------------------------------
// Row gather for downloads: dst[tl][w][:] = pool[loc[tl][w]][:]
// ---------------------------------------------------------------------------------------------
__global__ void k_gather_rows(const double* __restrict__ pool, const int32_t* __restrict__ loc,
                              double* __restrict__ dst, int64_t nrows, int D) {
    const int64_t total = nrows * D;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / D;
        const int d = (int)(i - r * D);
        dst[i] = pool[(size_t)loc[r] * D + d];
    }
}

__global__ void k_iota(int32_t* p, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        p[i] = (int32_t)i;
}

__global__ void k_tick(uint64_t* clock) { clock[0] += 1; }

// ---------------------------------------------------------------------------------------------
// Philox plan: random permutations by a bitonic sort of (random key | index) in LDS.
//   job j < n_split : split order of local rung j   -> order[j][W]
//   job j >= n_split: PT column permutation of rung (j - n_split) (global, 0..T-2) -> colslot[t][W]
//   the top rung's column map is the identity (written by job n_split + T - 1 without sorting).
// Draws depend only on (seed, iteration, purpose, global rung), so every rank of a sharded ladder
// builds the same PT plan.
// ---------------------------------------------------------------------------------------------
struct PlanArgs {
    int32_t* order;       // [NB][Tl][W]
    int32_t* colslot;     // [NB][T][W]
    const uint64_t* clock;
    uint64_t seed;
    int32_t Tl, T, W, NP2, rung_begin, n_split, jobs_per_iter, idx_bits;
};

__global__ __launch_bounds__(1024) void k_plan(const PlanArgs A) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    uint64_t* key = reinterpret_cast<uint64_t*>(smem_raw);
    const int tid = threadIdx.x;
    const int ib = blockIdx.x / A.jobs_per_iter;          // iteration within the batch
    const int job = blockIdx.x - ib * A.jobs_per_iter;
    const uint64_t it = A.clock[0] + (uint64_t)ib;
    const int W = A.W, NP2 = A.NP2;
    int32_t* out;
    uint32_t purpose, rung;
    if (job < A.n_split) {
        out = A.order + ((size_t)ib * A.Tl + job) * W;
        purpose = PURPOSE_SPLIT;
        rung = (uint32_t)(A.rung_begin + job);
    } else {
        const int t = job - A.n_split;
        out = A.colslot + ((size_t)ib * A.T + t) * W;
        purpose = PURPOSE_PTPERM;
        rung = (uint32_t)t;
        if (t == A.T - 1) {                                // identity for the hottest rung
            for (int i = tid; i < W; i += blockDim.x) out[i] = i;
            return;
        }
    }
    const uint64_t mask = (1ull << A.idx_bits) - 1ull;
    for (int i = tid; i < NP2; i += blockDim.x) {
        uint64_t kv;
        if (i < W) {
            const u4 ctr{(uint32_t)it, (uint32_t)(it >> 32), rung * (uint32_t)NP2 + (uint32_t)i, purpose};
            const u4 d = philox4x32_10(ctr, (uint32_t)A.seed, (uint32_t)(A.seed >> 32));
            kv = ((((uint64_t)d.x << 32) | d.y) & ~mask) | (uint64_t)i;
            kv &= ~(1ull << 63);                           // keep below the padding keys
        } else {
            kv = (1ull << 63) | (uint64_t)i;
        }
        key[i] = kv;
    }
    __syncthreads();
    for (int k = 2; k <= NP2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int idx = tid; idx < (NP2 >> 1); idx += blockDim.x) {
                const int i = ((idx & ~(j - 1)) << 1) | (idx & (j - 1));
                const int l = i | j;
                const bool asc = (i & k) == 0;
                const uint64_t a = key[i], b = key[l];
                if ((a > b) == asc) {
                    key[i] = b;
                    key[l] = a;
                }
            }
            __syncthreads();
        }
    }
    if (job >= A.n_split) {                                // PT column map: the permutation itself
        for (int i = tid; i < W; i += blockDim.x) out[i] = (int32_t)(key[i] & mask);
        return;
    }
    // Split order: the first N0 = ceil(W/2) entries of the permutation are the walkers of split 0
    // (a uniformly random balanced labelling, red_blue.py:119-124).  Emit each half in ASCENDING
    // walker order like the reference's boolean masks do: a tile of 64 moving walkers then touches
    // ~128 consecutive ids, so the per-walker scalars (loc, L, P, accept counts) are read and
    // written as whole cache lines instead of one line per 4..8-byte element.
    const int N0 = (W + 1) / 2;
    const int nt = blockDim.x;
    const int per = (NP2 + nt - 1) / nt;                   // permutation entries per thread
    uint32_t mine[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int i = tid + q * nt;
        mine[q] = (q < per && i < W) ? (uint32_t)(key[i] & mask) : 0xFFFFFFFFu;
    }
    __syncthreads();
    uint8_t* lab = reinterpret_cast<uint8_t*>(key);        // [W] labels, reusing the sort buffer
    uint32_t* scan = reinterpret_cast<uint32_t*>(lab + ((W + 15) & ~15));   // [nt] zero counts
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int i = tid + q * nt;
        if (q < per && i < W) lab[mine[q]] = (i >= N0) ? 1 : 0;
    }
    __syncthreads();
    const int chunk = (W + nt - 1) / nt;                   // consecutive ids per thread
    const int lo = tid * chunk, hi = min(W, lo + chunk);
    uint32_t z = 0;
    for (int i = lo; i < hi; ++i) z += (lab[i] == 0);
    scan[tid] = z;
    __syncthreads();
    for (int off = 1; off < nt; off <<= 1) {               // inclusive Hillis-Steele scan of zero counts
        const uint32_t v = (tid >= off) ? scan[tid - off] : 0u;
        __syncthreads();
        scan[tid] += v;
        __syncthreads();
    }
    uint32_t z0 = scan[tid] - z;                           // zeros before this thread's chunk
    uint32_t o0 = (uint32_t)lo - z0;                       // ones before it
    for (int i = lo; i < hi; ++i) {
        if (lab[i] == 0) out[z0++] = i;
        else out[N0 + o0++] = i;
    }
}

// ---------------------------------------------------------------------------------------------
// PT cascade in column form.
//
// The reference walks the pairs (i, i-1), i = T-1 .. 1, matching slot iperm_i[k] of rung i with
// slot i1perm_i[k] of rung i-1 (tempering.py:515-541).  Because every pair uses permutations, the
// element (i, k) depends on exactly one element of pair i+1 (the one whose cold slot is its hot
// slot).  The cascade is therefore W independent "columns", each visiting one slot per rung:
// the walker carried down the column is compared with the resident of the next rung; on a swap
// the resident moves up one rung and the carried walker keeps falling.  Columns are independent,
// so the whole T-1 step sequential cascade is one parallel kernel.
//
// k_pt_chain builds the column form from the reference's draws (parity mode); in Philox mode the
// plan kernel draws colslot directly (same distribution: independent uniform matchings).
// ---------------------------------------------------------------------------------------------
__global__ void k_pt_invert(const int64_t* __restrict__ iperm, int32_t* __restrict__ inv, int npairs, int W) {
    const int64_t n = (int64_t)npairs * W;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int j = (int)(i / W);
        inv[(size_t)j * W + iperm[i]] = (int32_t)(i - (int64_t)j * W);
    }
}

// thread c follows column c: rows j = 0..T-2 of iperm/i1perm are the pairs i = T-1-j.
__global__ void k_pt_chain(const int64_t* __restrict__ iperm, const int64_t* __restrict__ i1perm,
                           const int32_t* __restrict__ inv, const double* __restrict__ u_swap,
                           int32_t* __restrict__ colslot, int32_t* __restrict__ colk,
                           double* __restrict__ colu, int T, int W) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= W) return;
    int k = c;
    for (int j = 0; j < T - 1; ++j) {
        const int i = T - 1 - j;
        if (j == 0) colslot[(size_t)i * W + c] = (int32_t)iperm[k];
        const int b = (int)i1perm[(size_t)j * W + k];
        colslot[(size_t)(i - 1) * W + c] = b;
        colk[(size_t)j * W + c] = k;
        colu[(size_t)j * W + c] = u_swap[(size_t)j * W + k];
        if (j + 1 < T - 1) k = inv[(size_t)(j + 1) * W + b];
    }
}

struct PtArgs {
    const double* Lfull;        // [T][W] log-likelihood of the full ladder (== L when unsharded)
    const double* L;            // local current buffers [Tl][W]
    const double* P;
    const int32_t* loc;
    double* Lnew;               // local next buffers
    double* Pnew;
    int32_t* locnew;
    double* betas;              // [T], updated in place by the last block
    const int32_t* colslot;     // [T][W]
    const double* colu;         // [T-1][W] uniforms in column order (parity) or nullptr (Philox)
    uint8_t* selcol;            // [T-1][W] swap decisions in column order, row j <-> pair T-1-j (or nullptr)
    int32_t* srcfull;           // [T][W] global source slot id (t'*W + w') of the walker arriving at every slot of the full ladder (sharded) or nullptr
    unsigned long long* swap_part;   // [nblocks][T-1] per-workgroup swap counts, 8-byte granules
    double* swaps_last;         // [T-1]
    double* swaps_total;        // [T-1]
    unsigned* ticket;
    uint64_t* clock;            // iteration counter, advanced by the last block when tick != 0
    int64_t* adapt_time;
    uint64_t seed;
    double lag, nu;
    int64_t stop_adaptation;
    int32_t T, W, Tl, rung_begin, adapt, tick, sharded;
};

constexpr int PT_COLS = 16;      // columns per workgroup: W/16 workgroups keep every CU busy at W = 4096
constexpr int PT_THREADS = 256;

// LDS: Lc[T][PT_COLS] f64 column log-likelihoods, lu[T][PT_COLS] f64 log-uniforms, sbeta[T] f64,
//      src[T][PT_COLS] i16 rung each final slot takes its walker from, sel[T][PT_COLS] u8.
__host__ __device__ inline size_t pt_lds_layout(int T) { return (size_t)T * PT_COLS * (8 + 8 + 2 + 1) + (size_t)T * 8; }

template <bool PHILOX>
__global__ __launch_bounds__(PT_THREADS) void k_pt_cascade(const PtArgs A) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int T = A.T, W = A.W;
    double* Lc = reinterpret_cast<double*>(smem_raw);            // [T][PT_COLS]
    double* lu = Lc + (size_t)T * PT_COLS;                       // [T][PT_COLS] (row j = pair T-1-j)
    double* sbeta = lu + (size_t)T * PT_COLS;                    // [T]
    int16_t* src = reinterpret_cast<int16_t*>(sbeta + T);        // [T][PT_COLS]
    uint8_t* sel = reinterpret_cast<uint8_t*>(src + (size_t)T * PT_COLS);  // [T][PT_COLS]
    __shared__ int s_last;
    const int tid = threadIdx.x;
    const int c0 = blockIdx.x * PT_COLS;
    const uint64_t it = PHILOX ? A.clock[0] : 0;

    // phase 1: gather the column's log-likelihoods, log-uniforms and the ladder (independent loads)
    for (int t = tid; t < T; t += PT_THREADS) sbeta[t] = A.betas[t];
    for (int e = tid; e < T * PT_COLS; e += PT_THREADS) {
        const int t = e / PT_COLS, cc = e - t * PT_COLS, c = c0 + cc;
        if (c < W) {
            const int slot = A.colslot[(size_t)t * W + c];
            Lc[e] = A.Lfull[(size_t)t * W + slot];
            if (t < T - 1) {
                double u;
                if (PHILOX) {
                    const u4 ctr{(uint32_t)it, (uint32_t)(it >> 32), (uint32_t)(t * W + c), PURPOSE_PTU};
                    const u4 d = philox4x32_10(ctr, (uint32_t)A.seed, (uint32_t)(A.seed >> 32));
                    u = u01(d.x, d.y);
                } else {
                    u = A.colu[(size_t)t * W + c];
                }
                lu[e] = log(u);                                  // tempering.py:535
            }
        }
    }
    __syncthreads();

    // phase 2: one lane per column walks hot -> cold
    if (tid < PT_COLS && c0 + tid < W) {
        const int cc = tid;
        double cL = Lc[(size_t)(T - 1) * PT_COLS + cc];
        int cr = T - 1;                                          // rung the carried walker came from
        for (int i = T - 1; i >= 1; --i) {
            const int j = T - 1 - i;
            const double Lb = Lc[(size_t)(i - 1) * PT_COLS + cc];
            const double dbeta = sbeta[i - 1] - sbeta[i];        // tempering.py:518-522
            const double pacc = dbeta * (cL - Lb);               // tempering.py:538
            const bool s = pacc > lu[(size_t)j * PT_COLS + cc];  // tempering.py:541
            sel[(size_t)j * PT_COLS + cc] = s ? 1 : 0;
            if (s) {
                src[(size_t)i * PT_COLS + cc] = (int16_t)(i - 1);   // resident moves up, carried keeps falling
            } else {
                src[(size_t)i * PT_COLS + cc] = (int16_t)cr;        // carried walker settles on rung i
                cr = i - 1;
                cL = Lb;
            }
        }
        src[cc] = (int16_t)cr;
    }
    __syncthreads();

    // phase 3: scatter the permuted L / P / loc of the resident rungs; count swaps
    for (int e = tid; e < T * PT_COLS; e += PT_THREADS) {
        const int t = e / PT_COLS, cc = e - t * PT_COLS, c = c0 + cc;
        if (c >= W) continue;
        if (t < T - 1 && A.selcol) A.selcol[(size_t)t * W + c] = sel[e];
        const int st = src[e];
        const int dslot = A.colslot[(size_t)t * W + c];
        const int sslot = A.colslot[(size_t)st * W + c];
        if (A.srcfull) A.srcfull[(size_t)t * W + dslot] = st * W + sslot;
        const int tl = t - A.rung_begin;
        if (tl < 0 || tl >= A.Tl) continue;
        const size_t di = (size_t)tl * W + dslot;
        A.Lnew[di] = Lc[(size_t)st * PT_COLS + cc];
        const int stl = st - A.rung_begin;
        if (stl >= 0 && stl < A.Tl) {
            const size_t si = (size_t)stl * W + sslot;
            A.Pnew[di] = A.P[si];
            A.locnew[di] = A.loc[si];
        } else {
            A.locnew[di] = -1;       // row + log-prior arrive from another rank (hens_pt_finish_sharded)
        }
    }
    // Per-workgroup swap counts go to a private row with write-through (sc1) stores; the last
    // workgroup to take a ticket reduces them.  No release fence: a buffer_wbl2 per workgroup
    // writes back the whole XCD L2 and made this kernel 10x slower; device atomics on T-1
    // counters sharing one cache line were no better (MI355X_MICROARCH: publish forms R1/R2).
    for (int j = tid; j < T - 1; j += PT_THREADS) {
        unsigned n = 0;
        for (int cc = 0; cc < PT_COLS && c0 + cc < W; ++cc) n += sel[(size_t)j * PT_COLS + cc];
        __hip_atomic_store(&A.swap_part[(size_t)blockIdx.x * (T - 1) + (T - 2 - j)], (unsigned long long)n,
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // pair i = T-1-j -> index i-1
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // every storing wave drains
    __syncthreads();
    if (tid == 0)
        s_last = (__hip_atomic_fetch_add(A.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1) ? 1 : 0;
    __syncthreads();
    if (!s_last) return;
    // every thread of the last block sums a slice of the per-workgroup rows into LDS counters
    unsigned* cnt = reinterpret_cast<unsigned*>(lu);              // [T-1], lu is dead by now
    for (int j = tid; j < T - 1; j += PT_THREADS) cnt[j] = 0;
    __syncthreads();
    {
        // 8 independent agent-scope loads in flight per thread, then the LDS adds
        const unsigned total = gridDim.x * (unsigned)(T - 1);
        for (unsigned e0 = tid; e0 < total; e0 += 8 * PT_THREADS) {
            unsigned long long v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const unsigned e = e0 + q * PT_THREADS;
                v[q] = (e < total) ? __hip_atomic_load(&A.swap_part[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const unsigned e = e0 + q * PT_THREADS;
                if (v[q]) atomicAdd(&cnt[e % (unsigned)(T - 1)], (unsigned)v[q]);
            }
        }
    }
    __syncthreads();
    // Ladder adaptation, lane-parallel where the reference's arithmetic allows it (a single lane
    // running T divisions + exps serially cost ~18 us): ratios, dS, deltaT per lane; the cumsum stays
    // a sequential left-to-right sum like np.cumsum; the reciprocal and the update per lane again.
    double* r = Lc;                           // ratios [T-1]
    double* dT = Lc + T;                      // deltaTs, then their running sum [T-2]
    const bool do_adapt = A.adapt && T > 1;
    int64_t time = 0;
    bool moving = false;
    if (do_adapt) {
        time = A.adapt_time[0];
        moving = A.stop_adaptation < 0 || time < A.stop_adaptation;
    }
    for (int j = tid; j < T - 1; j += PT_THREADS) {
        const double n = (double)cnt[j];
        A.swaps_last[j] = n;
        A.swaps_total[j] += n;
        r[j] = n / (double)W;                                         // :587
    }
    __syncthreads();
    if (moving) {
        const double decay = A.lag / ((double)time + A.lag);          // :571
        const double kappa = decay / A.nu;                            // :572
        for (int j = tid; j + 2 < T; j += PT_THREADS) {
            const double dS = kappa * (r[j] - r[j + 1]);              // :575
            double d = 1.0 / sbeta[j + 1] - 1.0 / sbeta[j];           // :578
            d *= exp(dS);
            dT[j] = d;
        }
        __syncthreads();
        if (tid == 0)
            for (int j = 1; j + 2 < T; ++j) dT[j] = dT[j - 1] + dT[j];    // np.cumsum order
        __syncthreads();
        const double inv0 = 1.0 / sbeta[0];
        for (int j = tid; j + 2 < T; j += PT_THREADS) {
            const double bn = 1.0 / (dT[j] + inv0);                   // :580
            A.betas[j + 1] = sbeta[j + 1] + (bn - sbeta[j + 1]);      // :583,:593
        }
    }
    if (tid == 0) {
        if (do_adapt) A.adapt_time[0] = time + 1;                     // :596
        if (A.tick) A.clock[0] += 1;
        __hip_atomic_store(A.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// column-order decisions -> the reference's k order (row j, element colk[j][c])
__global__ void k_pt_sel_to_korder(const uint8_t* __restrict__ selcol, const int32_t* __restrict__ colk,
                                   uint8_t* __restrict__ selk, int npairs, int W) {
    const int64_t n = (int64_t)npairs * W;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int j = (int)(i / W);
        selk[(size_t)j * W + colk[i]] = selcol[i];
    }
}

// ---------------------------------------------------------------------------------------------
// Sharded ladder: rows that change rank.  Every rank replays the whole cascade from the
// all-gathered log-likelihoods, so every rank knows srcfull[t][w] = global slot the walker arriving
// at (t, w) comes from.  A row travels as D + 2 doubles: [destination global slot id (bit pattern) |
// x[0..D) | log-prior]; its log-likelihood is already in the gathered ladder.
// ---------------------------------------------------------------------------------------------
constexpr int MAX_RANKS = 16;

// counts[0..nranks) = rows this rank sends to each peer, counts[MAX_RANKS..) = rows it receives
__global__ void k_xchg_count(const int32_t* __restrict__ srcfull, const int32_t* __restrict__ rank_of_rung,
                             int T, int W, int me, unsigned* __restrict__ counts) {
    __shared__ unsigned s_cnt[2 * MAX_RANKS];
    if (threadIdx.x < 2 * MAX_RANKS) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    const int64_t n = (int64_t)T * W;
    for (int64_t g = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; g < n; g += (int64_t)gridDim.x * blockDim.x) {
        const int src = srcfull[g];
        const int dr = rank_of_rung[g / W], sr = rank_of_rung[src / W];
        if (dr != sr) {
            if (sr == me) atomicAdd(&s_cnt[dr], 1u);
            if (dr == me) atomicAdd(&s_cnt[MAX_RANKS + sr], 1u);
        }
    }
    __syncthreads();
    if (threadIdx.x < 2 * MAX_RANKS && s_cnt[threadIdx.x]) atomicAdd(&counts[threadIdx.x], s_cnt[threadIdx.x]);
}

// cursors[p] starts at the offset of peer p's segment in the send buffer
__global__ void k_xchg_fill(const int32_t* __restrict__ srcfull, const int32_t* __restrict__ rank_of_rung,
                            int T, int W, int me, int rung_begin, unsigned* __restrict__ cursors,
                            int32_t* __restrict__ send_slot, int32_t* __restrict__ send_dest) {
    const int64_t n = (int64_t)T * W;
    for (int64_t g = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; g < n; g += (int64_t)gridDim.x * blockDim.x) {
        const int src = srcfull[g];
        const int dr = rank_of_rung[g / W], sr = rank_of_rung[src / W];
        if (dr != sr && sr == me) {
            const unsigned pos = atomicAdd(&cursors[dr], 1u);
            send_slot[pos] = src - rung_begin * W;       // local slot of the leaving walker
            send_dest[pos] = (int32_t)g;
        }
    }
}

__global__ void k_pack_rows(const double* __restrict__ pool, const int32_t* __restrict__ loc,
                            const double* __restrict__ P, const int32_t* __restrict__ send_slot,
                            const int32_t* __restrict__ send_dest, double* __restrict__ out, int64_t nsend, int D) {
    const int64_t total = nsend * (D + 2);
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t e = i / (D + 2);
        const int d = (int)(i - e * (D + 2));
        const int slot = send_slot[e];
        double v;
        if (d == 0) v = __longlong_as_double((long long)send_dest[e]);
        else if (d <= D) v = pool[(size_t)loc[slot] * D + (d - 1)];
        else v = P[slot];
        out[i] = v;
    }
}

__global__ void k_unpack_rows(double* __restrict__ pool, int32_t* __restrict__ locnew, double* __restrict__ Pnew,
                              const double* __restrict__ in, int64_t nrecv, int D, int W, int rung_begin,
                              int32_t free_off) {
    const int64_t total = nrecv * (D + 2);
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t e = i / (D + 2);
        const int d = (int)(i - e * (D + 2));
        const int slot = (int)__double_as_longlong(in[e * (D + 2)]) - rung_begin * W;
        if (d == 0) locnew[slot] = free_off + slot;
        else if (d <= D) pool[(size_t)(free_off + slot) * D + (d - 1)] = in[i];
        else Pnew[slot] = in[i];
    }
}

}  // namespace hens
