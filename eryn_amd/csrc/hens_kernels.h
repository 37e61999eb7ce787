// hens_kernels.h - gfx950 (CDNA4, wave64) device code of libhipensemble.
//
// Compiled with -ffp-contract=off: every a*b+c below rounds twice like NumPy unless it is
// written as an explicit fma().  That is what makes the proposal q = c - (c - s) * zz
// (stretch.py:143-145) and the accept arithmetic (red_blue.py:292) bit-identical to the
// reference.  All file:line citations are relative to /root/reference/src/eryn.
//
// Data layout in HBM (one context = the ladder shard [rung_begin, rung_end)):
//   pool   f64 [2 * Tl * W][D]   walker rows, AoS (a row is one contiguous 8*D-byte burst).
//                                Every walker (tl, w) owns two "home" rows, tl*W+w and
//                                Tl*W + tl*W+w; iteration parity p writes into home_p and reads
//                                through `loc`, so a stretch step is read-own + read-complement +
//                                write-own with no in-place hazard, and the PT cascade never moves
//                                a row: it permutes `loc`.
//   loc    i32 [Tl * W]          pool row currently holding walker (tl, w)   (double buffered)
//   L, P   f64 [Tl * W]          log-likelihood / log-prior                  (double buffered)
//   betas  f64 [T]               full ladder                                  (double buffered)
//   draws  per iteration, per rung, per split position: own i32, cw i32, zz f64, fac f64, lu f64
//          - everything about a proposal that does not depend on the state.  Built ahead of time
//          by the plan kernel (Philox) or from the caller's NumPy draws (parity mode).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

namespace hens {

constexpr int TILE = 64;                    // walkers per workgroup = one wavefront of walker-lanes
constexpr unsigned FLAG_NONFINITE_X = 1u;   // inf/NaN coordinate seen (ensemble.py:1258-1262)
constexpr unsigned FLAG_NAN_LOGL = 2u;      // NaN likelihood (red_blue.py:279-281)

constexpr unsigned FLAG_PIPE_TIMEOUT = 4u; // ladder pipeline: a neighbour's flag did not arrive in time

// A/B knobs of dev builds (tools/devbuild.sh -D...)
enum { LIKE_DENSE = 0, LIKE_DIAG = 1, LIKE_ROSEN = 2, LIKE_HOST = 3 };
// what a stretch-kernel launch does: a red/blue stretch half-step, the evaluation of the resident state, or a
// full-ensemble Metropolis-Hastings proposal q = x + step (mh.py:56-193; the step rows are read where the
// stretch move reads the complement walker)
enum { MODE_STRETCH = 0, MODE_EVAL = 1, MODE_MH = 2 };

// ---------------------------------------------------------------------------------------------
// Philox4x32-10 (Salmon et al. 2011), counter-based: draws are a pure function of
// (seed, iteration, purpose, global rung, walker), so any kernel / any rank regenerates them.
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
// Keyed pseudo-random permutation of [0, W): an 8-round alternating Feistel network on idx_bits bits, cycle-walked into
// [0, W).  A bijection for every key, no memory, no sort: position c of rung t's permutation is prp(c) wherever it is
// needed, and every rank of a sharded ladder computes the same value.  Round function (round 3): two 24-bit multiplies
// with a shift-xor between them - v_mul_u32_u24 issues at full rate, the 32-bit multiplies of the murmur3 finaliser used
// before at a quarter of it, and this network sits on the critical path of every launch that maps columns to walkers
// (half-blocks have at most 11 bits; the round key supplies the rest of the 24).  tools/rng_validate.py: first- and
// second-order uniformity of the permutations as good as with the finaliser, also at 6 rounds; 8 are kept.
struct PrpKey { uint32_t k[8]; };
__device__ __forceinline__ uint32_t fmix32(uint32_t h) {
    h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
    return h;
}
__device__ __forceinline__ uint32_t pmix(uint32_t x, uint32_t k) {
    uint32_t h = __umul24(x ^ k, 0x9E3779u);              // (the low 24 bits of either operand)
    h ^= h >> 15;
    h = __umul24(h, 0x85EBCBu);
    return h >> 11;
}
__device__ __forceinline__ PrpKey prp_key(uint64_t seed, uint64_t it, uint32_t purpose, uint32_t rung) {
    const u4 a = philox4x32_10(u4{(uint32_t)it, (uint32_t)(it >> 32), rung, purpose}, (uint32_t)seed, (uint32_t)(seed >> 32));
    const u4 b = philox4x32_10(u4{(uint32_t)it, (uint32_t)(it >> 32), rung, purpose ^ 0x100u}, (uint32_t)seed, (uint32_t)(seed >> 32));
    return PrpKey{{a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w}};
}
__device__ __forceinline__ uint32_t prp(uint32_t x, const uint32_t* k, int bits, uint32_t W) {
    const int lb = bits >> 1, rb = bits - lb;            // left (high) / right (low) half widths
    const uint32_t lm = (1u << lb) - 1u, rm = (1u << rb) - 1u;
    do {
        uint32_t L = x >> rb, R = x & rm;
#pragma unroll
        for (int r = 0; r < 8; r += 2) {
            L ^= pmix(R, k[r]) & lm;
            R ^= pmix(L, k[r + 1]) & rm;
        }
        x = (L << rb) | R;
    } while (x >= W);                                     // cycle walking keeps it a bijection on [0, W)
    return x;
}

// inverse of prp: walk the cycle backwards until the value is inside [0, W) again
__device__ __forceinline__ uint32_t prp_inv(uint32_t y, const uint32_t* k, int bits, uint32_t W) {
    const int lb = bits >> 1, rb = bits - lb;
    const uint32_t lm = (1u << lb) - 1u, rm = (1u << rb) - 1u;
    do {
        uint32_t L = y >> rb, R = y & rm;
#pragma unroll
        for (int r = 6; r >= 0; r -= 2) {
            R ^= pmix(L, k[r + 1]) & rm;
            L ^= pmix(R, k[r]) & lm;
        }
        y = (L << rb) | R;
    } while (y >= W);
    return y;
}

// (the PT helpers that use these follow the enum)
enum : uint32_t { PURPOSE_STRETCH = 0, PURPOSE_STRETCH_ACC = 2, PURPOSE_SPLIT = 8, PURPOSE_PTPERM = 9,
                  PURPOSE_PTU = 10, PURPOSE_MH_ACC = 11, PURPOSE_MH_NORMAL = 12, PURPOSE_MOVE = 13 };
enum { MH_ISO = 0, MH_DIAG = 1, MH_FULL = 2 };

// Philox-mode PT draws: column c of the cascade meets slot pt_slot(t, c) of global rung t - every rung its own keyed
// permutation, i.e. pair (i, i-1) is matched through two permutations like the reference's (tempering.py:526-532) - and
// pair (i, i-1) on column c consumes the uniform of row j = T-1-i.
__device__ __forceinline__ int pt_slot(uint64_t seed, uint64_t it, int t, int T, int c, int idx_bits, int W) {
    const PrpKey K = prp_key(seed, it, PURPOSE_PTPERM, (uint32_t)t);
    return (int)prp((uint32_t)c, K.k, idx_bits, (uint32_t)W);
}
// Block-balanced split labelling (Philox mode, tempered ladders of up to 64 rungs).  The cascade's columns are cut into
// blocks of cb consecutive columns; on every rung the walkers met by the first cb/2 columns of a block move in the first
// half-step, those met by the other cb/2 in the second.  Because a rung's column map is a uniform permutation, this is a
// uniformly random balanced labelling of the rung like the reference's shuffle of arange(W) % 2 (red_blue.py:119-124) -
// and one workgroup that owns a block of columns owns exactly 64 = (cb/2) * T walkers of the second half-step plus
// everything the cascade of those columns touches, so the second half-step and the cascade run as ONE launch
// (k_split1_pt).  A half is enumerated by PLACE p = block * cb/2 + member; the walker at place p of half h on rung t is
// prp_t(place_column(h, p)) - a closed form, so whoever needs a walker's complement computes it in registers (round 2
// ranked per-block hashes: a uniform place of the other half then needed a table of the whole rung, i.e. a plan kernel).
// Split and matching of one iteration come from the same permutation (each uniform on its own; any state-independent
// choice of partition and matching leaves the chain's stationary distribution untouched).
__device__ __forceinline__ int place_column(int h, int p, int hb_shift) {      // hb = cb / 2 = 1 << hb_shift
    return ((p >> hb_shift) << (hb_shift + 1)) + (h << hb_shift) + (p & ((1 << hb_shift) - 1));
}
__device__ __forceinline__ double pt_uniform(uint64_t seed, uint64_t it, int j, int W, int c) {   // tempering.py:535
    const u4 ctr{(uint32_t)it, (uint32_t)(it >> 32), (uint32_t)(j * W + c), PURPOSE_PTU};
    const u4 d = philox4x32_10(ctr, (uint32_t)seed, (uint32_t)(seed >> 32));
    return u01(d.x, d.y);
}

// Everything random about one stretch proposal from ONE Philox4x32 call keyed by (iteration, global rung, walker): the two
// 53-bit uniforms behind zz (stretch.py:129-132) and the accept test (red_blue.py:294), and - from the 2 x 11 low bits the
// uniforms do not use - a 22-bit number that indexes the complement half (stretch.py:93-99; a second call for the accept
// uniform alone was a quarter of the plan's instructions).
struct StretchDraw { double uz, ua; uint32_t r22; };
constexpr int STRETCH_INDEX_BITS = 22;
__device__ __forceinline__ StretchDraw stretch_draw(uint64_t seed, uint64_t it, uint32_t wid) {
    const u4 d = philox4x32_10(u4{(uint32_t)it, (uint32_t)(it >> 32), wid, PURPOSE_STRETCH}, (uint32_t)seed, (uint32_t)(seed >> 32));
    return StretchDraw{u01(d.x, d.y), u01(d.z, d.w), ((d.y & 0x7FFu) << 11) | (d.w & 0x7FFu)};
}
__device__ __forceinline__ int stretch_index(uint32_t r22, int Nc) {          // uniform on [0, Nc), Nc <= 2^22
    return (int)(((uint64_t)r22 * (uint64_t)Nc) >> STRETCH_INDEX_BITS);
}

// Everything about split position q = h N0 + p of rung `rung` in iteration `it`: the walker there, its complement - the
// walker at a uniform place of the other half - and the two uniforms (one Philox call keyed by the POSITION, so it does
// not wait for the round keys).
struct PlaceDraw { int own, cw; double uz, ua; int r; };
__device__ __forceinline__ PlaceDraw stretch_draws_at(uint64_t seed, uint64_t it, uint32_t rung, int q, const uint32_t* key,
                                                      int W, int idx_bits, int hb_shift) {
    const int N0 = W >> 1;
    const int h = q >= N0 ? 1 : 0, p = q - h * N0;
    const StretchDraw sd = stretch_draw(seed, it, rung * (uint32_t)W + (uint32_t)q);
    const int r = stretch_index(sd.r22, N0);
    const int own = (int)prp((uint32_t)place_column(h, p, hb_shift), key, idx_bits, (uint32_t)W);
    const int cw = (int)prp((uint32_t)place_column(1 - h, r, hb_shift), key, idx_bits, (uint32_t)W);
    return PlaceDraw{own, cw, sd.uz, sd.ua, r};
}

// One Box-Muller pair of standard normals for coordinates (2 pr, 2 pr + 1) of walker `wid` (= rung * W + walker)
// in iteration `it`: the draw of the Gaussian MH move (k_mh_draw and the inline MODE_MH path share it).
// A Box-Muller pair of standard normals out of TWO 32-bit words (round 5; rounds 3-4 spent a whole Philox result - two 53-bit
// uniforms - on a pair whose transcendental part runs in single precision anyway).  The transcendental part on the hardware units
// (v_log_f32, v_sin_f32 / v_cos_f32 take the angle in revolutions): a proposal step needs the N(0, 1) shape, not 53 bits - FP64
// log / sqrt / sincospi made a launch that draws its normals in place ALU-bound.  The radius' uniform is (a + 1) / 2^32 in (0, 1]
// (largest radius sqrt(64 ln 2) = 6.66); the proposal stays symmetric (cos / sin of a uniform angle), which is all detailed
// balance asks of it.
__device__ __forceinline__ double2 mh_normal_from32(const uint32_t a, const uint32_t b) {
    const float u1 = ((float)a + 1.0f) * 2.3283064365386963e-10f;             // (0, 1]
    const float ang = (float)b * 2.3283064365386963e-10f;                     // revolutions
    const float lf = __builtin_amdgcn_logf(u1) * 0.69314718056f;              // ln u1 <= 0
    const float r = __builtin_sqrtf(-2.0f * lf);
    return double2{(double)(r * __builtin_amdgcn_cosf(ang)), (double)(r * __builtin_amdgcn_sinf(ang))};
}
__device__ __forceinline__ uint32_t mh_normal_key(uint32_t pr) { return PURPOSE_MH_NORMAL | (pr << 8); }
// FOUR normals per Philox call: the pairs (2 pr, 2 pr + 1) of TWO walkers of a rung, g rows apart (w with bit g clear: words x, y;
// bit g set: z, w) - the two rows one lane of the in-place MH launch owns in adjacent passes (g = its rows per pass, mh_pair_rows(D);
// g = 0: every walker its own call, words x, y).  The launch was bound by the 40 quarter-rate multiplies of a call per pair
// (config 5 on one GPU: 147 us for 268 MB of rows); k_mh_draw and hens_debug_draws evaluate the same function of (walker, pair).
__device__ __forceinline__ u4 mh_normal_quad(uint64_t seed, uint64_t it, uint32_t rung, uint32_t W, uint32_t w_low, uint32_t pr) {
    const u4 ctr{(uint32_t)it, (uint32_t)(it >> 32), rung * W + w_low, mh_normal_key(pr)};
    return philox4x32_10(ctr, (uint32_t)seed, (uint32_t)(seed >> 32));
}
__device__ __forceinline__ double2 mh_normal_pair(uint64_t seed, uint64_t it, uint32_t rung, uint32_t W, uint32_t w, uint32_t pr, uint32_t g) {
    const u4 d = mh_normal_quad(seed, it, rung, W, w & ~g, pr);
    return (w & g) ? mh_normal_from32(d.z, d.w) : mh_normal_from32(d.x, d.y);
}
__host__ __device__ constexpr int mh_pair_rows(int D) {        // rows per pass of k_stretch_fast<D> (RPP) where a lane owns two rows or more
    return D == 128 ? 8 : (D == 64 ? 16 : ((D == 32 || D == 16) ? 32 : 0));
}
__device__ __forceinline__ double mh_uniform(uint64_t seed, uint64_t it, uint32_t wid) {         // mh.py:157
    const u4 ctr{(uint32_t)it, (uint32_t)(it >> 32), wid, PURPOSE_MH_ACC};
    const u4 d = philox4x32_10(ctr, (uint32_t)seed, (uint32_t)(seed >> 32));
    return u01(d.x, d.y);
}
__device__ __forceinline__ double mh_log_uniform(uint64_t seed, uint64_t it, uint32_t wid) { return log(mh_uniform(seed, it, wid)); }

// ---------------------------------------------------------------------------------------------
// Ladder-pipeline primitives (used by the stretch kernels too; the protocol is described at k_pipe_*).
// ---------------------------------------------------------------------------------------------
// mailbox flag words (sweep counters raised by the peers)
enum { PF_LUP = 0, PF_LDN = 1, PF_ROWS_TOP = 2, PF_CNT0 = 8, PIPE_FLAG_WORDS = 64 };

// one rank's mailbox (uncached device memory, peer-mapped through HIP IPC); the protocol is described at k_pipe_*
struct PipeBox {
    unsigned* flags;       // [PIPE_FLAG_WORDS] sweep counters raised by the peers
    long long* meta;       // [32]: [par] = pool row of slot 0 of the cold neighbour's hottest rung after its stretch move
    unsigned* lupf;        // [W / PT_COLS] sweep counters: block b of the hot neighbour's walk has stored its columns' lp_up
    unsigned* counts;      // [4][T] accepted swaps per pair (index i-1 for pair (i, i-1)), written by the pair's owner; buffer = sweep & 3
    double* lp_up;         // [2][2][W] (L, P) carried by each column of the hot neighbour (column order)
    double* lp_dn;         // [2][2][W] (L, P) of the cold neighbour's hottest rung after its stretch move (slot order)
    double* guest;         // [2][2][W][D] arrived rows: side 0 = from the hot neighbour, side 1 = from the cold one
    // fused pipeline iteration (k_split1_pt<PIPE>, round 3): hand-offs per COLUMN BLOCK of the fused launch (128 / Tl columns)
    unsigned* blk_lup;     // [W / 2 + 1] sweep counters: block b of the hot neighbour's launch has stored its columns' lp_up
    unsigned* blk_ldn;     // [W / 2 + 1] block b of the cold neighbour's launch has published its hottest rung (lp_dn, ldn_loc, rows)
    unsigned* blk_rows;    // [W / 2 + 1] block b of the hot neighbour's launch has pushed its rows into my guest area
    int32_t* ldn_loc;      // [2][W] pool row of every walker of the cold neighbour's hottest rung (slot order): rows are
                           // updated in place there, so "where its rows are" is no longer a formula
};
__host__ __device__ inline size_t pipe_round(size_t n) { return (n + 255) & ~(size_t)255; }
__host__ __device__ inline size_t pipe_box_bytes(int T, int W, int D) {
    return pipe_round(PIPE_FLAG_WORDS * 4) + 256 + pipe_round(((size_t)W / 16 + 1) * 4) + pipe_round((size_t)4 * T * 4) +
           2 * pipe_round((size_t)4 * W * 8) +
           pipe_round((size_t)4 * W * D * 8) + 3 * pipe_round(((size_t)W / 2 + 1) * 4) + pipe_round((size_t)2 * W * 4);
}
__host__ __device__ inline PipeBox pipe_box(char* base, int T, int W, int D) {
    PipeBox b;
    size_t off = 0;
    b.flags = reinterpret_cast<unsigned*>(base + off); off += pipe_round(PIPE_FLAG_WORDS * 4);
    b.meta = reinterpret_cast<long long*>(base + off); off += 256;
    b.lupf = reinterpret_cast<unsigned*>(base + off); off += pipe_round(((size_t)W / 16 + 1) * 4);
    b.counts = reinterpret_cast<unsigned*>(base + off); off += pipe_round((size_t)4 * T * 4);
    b.lp_up = reinterpret_cast<double*>(base + off); off += pipe_round((size_t)4 * W * 8);
    b.lp_dn = reinterpret_cast<double*>(base + off); off += pipe_round((size_t)4 * W * 8);
    b.guest = reinterpret_cast<double*>(base + off); off += pipe_round((size_t)4 * W * D * 8);
    b.blk_lup = reinterpret_cast<unsigned*>(base + off); off += pipe_round(((size_t)W / 2 + 1) * 4);
    b.blk_ldn = reinterpret_cast<unsigned*>(base + off); off += pipe_round(((size_t)W / 2 + 1) * 4);
    b.blk_rows = reinterpret_cast<unsigned*>(base + off); off += pipe_round(((size_t)W / 2 + 1) * 4);
    b.ldn_loc = reinterpret_cast<int32_t*>(base + off);
    return b;
}
constexpr int32_t PIPE_NOSEL = INT32_MIN;      // bottom boundary: the column's pair does not swap
// guest row index of column c (sweep parity par, side) and its `loc` encoding
__host__ __device__ inline int32_t pipe_guest_loc(int par, int side, int W, int c) { return ~((par * 2 + side) * W + c); }

// peer memory is written and read with system-scope accesses (sc0 sc1: nothing lingers in a cache)
// debug stamps (hens_debug_trace): the shader clock (s_memtime: per XCD, not synchronised) or - compiled with
// -DHENS_TRACE_REALTIME, tools/trace_skew.py - the 100 MHz wall clock, which all XCDs share
__device__ __forceinline__ unsigned long long trace_stamp() {
#ifdef HENS_TRACE_REALTIME
    return (unsigned long long)wall_clock64();
#else
    return __builtin_amdgcn_s_memtime();
#endif
}
// A kernel argument read where it is USED, not at the kernel's entry: the compiler hoists every by-value argument's scalar load to
// the top (it is invariant and dereferenceable), and an argument that is first needed behind the likelihood lives in SGPRs across
// it - k_split1_pt<32> spilled 53 of them into vector lanes (v_writelane / v_readlane, 176 such instructions in the kernel).  The
// opaque offset keeps this load at its program point.
template <class Tp>
__device__ __forceinline__ Tp late_kernarg(size_t off) {
    asm volatile("" : "+s"(off));
    const __attribute__((address_space(4))) char* base = (const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr();
    return *reinterpret_cast<const __attribute__((address_space(4))) Tp*>(base + off);
}

__device__ __forceinline__ void sys_store(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ double sys_load(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }

// Write-through stores (agent scope, `sc1`): what the NEXT launch of a call reads - on whichever XCD - is written through to memory
// by the two stepping launches of one GPU, so no line of it sits dirty in an XCD's L2 when the launch ends; every wave waits for its
// stores' acknowledgements before it ends (launch_end_wait), and the launch's AQL packet carries NO release fence: the packet
// processor's end-of-kernel L2 write-back cost 1.4 us per launch at config 2 with nothing to write back (round 5, LABNOTES 10.13).
// The acquire fence at the head of the next launch stays (it drops the clean copies).
__device__ __forceinline__ void wt_store(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void wt_store(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void wt_store(int32_t* p, int32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void launch_end_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }


// Flag discipline: everything a peer reads is stored with system-scope write-through stores (sys_store),
// so raising a flag only needs those stores COMPLETE (s_waitcnt vmcnt(0)), not an L2 write-back: the
// compiler's system-scope release would write back the whole L2 - megabytes of dirty walker rows that
// no peer ever reads - and costs ~5 us per flag.
// Release side: the asm wait (with its memory clobber: also a compiler barrier) holds the flag back until every
// earlier store of this wave has been ACKNOWLEDGED; a write-through system-scope store is acknowledged by the memory it
// targets (the peer's HBM over xGMI), i.e. when it is visible to a cache-bypassing reader there.  The flag itself is a
// system-scope store behind that wait.  (Assumption stated in DESIGN.md 6.1; the self-test hens_pipe_selftest checks
// the three access patterns on the actual node before the pipeline is used.)
__device__ __forceinline__ void pipe_raise(unsigned* f, uint32_t v) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_store(f, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Ticket: a workgroup whose own stores have completed takes a number; returns true in the workgroup that completes
// the cumulative target (launches of different grid sizes share one ticket) - it may then raise flags on behalf of
// the whole grid (no extra launch, no L2 write-back: the data the flags cover was written with write-through stores).
__device__ __forceinline__ bool pipe_last_ticket(unsigned* ticket, uint32_t target) {
    __shared__ int s_last_t;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) s_last_t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u == target;
    __syncthreads();
    return s_last_t != 0;
}

// A flag wait's budget is given in ticks of the 100 MHz wall clock but measured on the shader clock (s_memtime, a scalar load's
// worth of cycles): reading the wall clock (s_memrealtime) takes ~1.5 us on this chip, and rounds 2-4 read it once in front of
// every spin that had to wait at all and once per poll - a wait for a value one memory round trip away cost 3 us and more,
// every wait was quantised to 1.5 us.  The shader clock runs at <= 2.4 GHz: a budget of b ticks is at least b * 10 ns.
__device__ __forceinline__ bool spin_expired(unsigned long long t0, long long budget_ticks) {
    return (long long)(__builtin_amdgcn_s_memtime() - t0) > budget_ticks * 24;
}

// a workgroup's thread 0 spins until flag >= target (or the budget runs out: a peer died - fail the run
// instead of hanging the GPU); callers follow with __syncthreads()
// inject (dev hook, HENS_PIPE_INJECT_CYCLES): the flag counts as raised only once the shader clock has passed `inject_until` - a
// neighbour whose message arrives that late (see StretchArgs::inject_cycles)
__device__ __forceinline__ void pipe_spin(const unsigned* f, uint32_t target, long long budget, unsigned* err,
                                          unsigned long long* stats = nullptr, unsigned long long inject_until = 0) {
    if (!stats && !inject_until && __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) >= target) {
        __atomic_signal_fence(__ATOMIC_SEQ_CST);
        return;
    }
    // (a wait of this context has timed out before: the run has failed - every later wait of its queued launches would spend the
    //  whole budget again, 8 ranks x iterations x 2 launches of them one after the other; slow path only)
    if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & FLAG_PIPE_TIMEOUT) return;
    const long long w0 = stats ? wall_clock64() : 0;      // (debug statistics only: ~1.5 us per reading)
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < target ||
           (inject_until && (long long)(__builtin_amdgcn_s_memtime() - inject_until) < 0)) {
        __builtin_amdgcn_s_sleep(2);
        if (spin_expired(t0, budget)) {
            atomicOr(err, FLAG_PIPE_TIMEOUT);
            return;
        }
    }
    if (stats) {                       // debug (HENS_PIPE_STATS): ticks spent waiting and number of waits at this site
        atomicAdd(stats, (unsigned long long)(wall_clock64() - w0));
        atomicAdd(stats + 1, 1ull);
    }
    // Acquire side of the hand-off.  Hardware: the wave has waited for the flag's value (the branch depends on it), and
    // everything it reads from a peer afterwards is a system-scope (sc0 sc1) load that no cache serves, issued after
    // this point in program order.  Compiler: nothing may be moved above the spin.
    __atomic_signal_fence(__ATOMIC_SEQ_CST);
}

// Arrive / collect: every workgroup, once its own stores have completed, adds itself to the ticket and leaves
// (a no-return atomic: nothing to wait for); workgroup 0 - dispatched first, so always resident - polls the ticket
// until the whole grid has arrived and then acts for the launch.  Cheaper than "the last arriver acts": no
// workgroup pays the round trip of a returning atomic on its way out.
__device__ __forceinline__ bool pipe_arrive_collect(unsigned* ticket, unsigned nblocks, uint32_t sweep, long long budget,
                                                    unsigned* err) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) (void)__hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (blockIdx.x != 0) return false;
    if (threadIdx.x == 0 && __hip_atomic_load(ticket, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != nblocks * (sweep + 1u)) {
        const unsigned target = nblocks * (sweep + 1u);              // cumulative over the sweeps (mod 2^32)
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        while (__hip_atomic_load(ticket, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != target) {
            __builtin_amdgcn_s_sleep(1);
            if (spin_expired(t0, budget)) {
                atomicOr(err, FLAG_PIPE_TIMEOUT);
                break;
            }
        }
    }
    __syncthreads();
    return true;
}

// loc >= 0: a row of the pool.  loc < 0: guest row ~loc of this rank's mailbox (a walker that arrived
// through the ladder pipeline during the last PT sweep); `guest_delta` = (guest - pool) in doubles.
// (branch-free, round 5: as `loc >= 0 ? ... : ...` every use compiled to a divergent branch with the reload of a spilled SGPR block
//  - guest_delta's - inside it, and the wait for the LDS read of `loc` in front of it: a pipeline rank's gather passes issued their
//  row requests 40 - 55 instructions apart where one GPU's go out back to back)
__device__ __forceinline__ int64_t row_off(int32_t loc, int D, int64_t guest_delta) {
    const int32_t sg = loc >> 31;                                    // 0 / -1
    return (int64_t)(loc ^ sg) * D + ((int64_t)sg & guest_delta);    // loc ^ -1 = ~loc
}

// All rows of one swap-count accumulation buffer of k_split1_pt<PIPE> (nrows = 8 G rows of np pairs, G in {1, 2, 4, 8}, np <= 64 / G:
// pipe_acc_rows / acc_row_groups) summed by ONE wavefront in ONE memory round trip: lane = (row group g, pair p), 8 loads in flight
// per lane, the groups added with lane exchanges.  Every lane returns the total of pair lane % (64 / G) (pairs >= np: 0).
// (Round 5: rounds 3-4 summed the rows in nrows / 8 dependent batches of 8 loads on np lanes - under the row gathers of a launch in
//  full swing every batch is a 1.5-2 us round trip, and the adapting wave of a pipeline rank reached its first barrier 12.7 us into a
//  22 us launch: tools/pipe_trace.py, LABNOTES 10.)
__device__ __forceinline__ unsigned acc_rows_sum(const uint32_t* rows, int nrows, int np, int lane) {
    const int G = nrows >> 3, P2 = 64 / G, p = lane & (P2 - 1), g = lane / P2;
    unsigned u[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) u[r] = (p < np) ? rows[(size_t)(r * G + g) * np + p] : 0u;
    unsigned sum = 0;
#pragma unroll
    for (int r = 0; r < 8; ++r) sum += u[r];
    for (int m = P2; m < 64; m <<= 1) sum += __shfl_xor(sum, m);
    return sum;
}
__device__ __forceinline__ void acc_rows_clear(uint32_t* rows, int nrows, int np, int lane) {
    for (int e = lane; e < nrows * np; e += 64) rows[e] = 0u;
}

// One wavefront sums the swap counts a launch of k_split1_pt<PIPE> accumulated (acc_rows_sum), clears them if asked (sole reader;
// a caller with a second reader clears later: k_stretch_fast) and publishes the sums to every rank's mailbox: counts of sweep
// `sweep`, flag PF_CNT0 + rank.
__device__ __forceinline__ void pipe_push_counts(const uint32_t* rows_c, int nrows, int np, char* const* boxes, int nranks, int rank,
                                                 int T, int rung_begin, int W, int D, uint32_t sweep, int lane, bool clear_now = true) {
    const unsigned sum = acc_rows_sum(rows_c, nrows, np, lane);
    if (clear_now) acc_rows_clear(const_cast<uint32_t*>(rows_c), nrows, np, lane);
    if (lane < np)                                             // (group 0: lane = pair; local pair j+1 = global pair (rung_begin+j+1, rung_begin+j))
        for (int q = 0; q < nranks; ++q)
            __hip_atomic_store(pipe_box(boxes[q], T, W, D).counts + (size_t)(sweep & 3u) * T + (rung_begin + lane), sum, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_SYSTEM);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane < nranks) pipe_raise(pipe_box(boxes[lane], T, W, D).flags + PF_CNT0 + rank, sweep + 1);
    // (the flag stores are left in flight: nothing here depends on them, and the launch does not end before they have landed)
}

// Periodic parameters (utils/periodic.py, used by stretch.py:136-154 and gaussian.py:110-115); period <= 0: not periodic.
// distance(p1 = s, p2 = c): c - s, measured the short way round when it exceeds half a period (periodic.py:80-113: the moving
// point is shifted by one period towards the complement - new_s = -(period - s) or period + s - and the difference taken
// again); wrap: NumPy's float remainder (periodic.py:143-145; npy_divmod: fmod, shifted by the divisor when the signs
// differ, +0 of the divisor's sign when exact).  fmod is exact, so both reproduce the reference bit for bit.
__device__ __forceinline__ double periodic_diff(double s, double c, double period) {
    double diff = c - s;
    if (period > 0.0 && fabs(diff) > period / 2.0) {
        const double new_s = diff < 0.0 ? -(period - s) : period + s;
        diff = c - new_s;
    }
    return diff;
}
__device__ __forceinline__ double periodic_wrap(double q, double period) {
    if (!(period > 0.0)) return q;
    double m = fmod(q, period);
    if (m != 0.0) {
        if (m < 0.0) m += period;
    } else {
        m = 0.0;                                  // copysign(0, period), period > 0
    }
    return m;
}

// The state-independent part of one proposal (stretch.py:129-132,223; red_blue.py:294).
struct Draws {
    int32_t* own;    // [Tl][W] moving walker at each split position (positions < N0: split 0)
    int32_t* cw;     // [Tl][W] its complement walker
    double* zz;      // [Tl][W] stretch factor ((a-1) u + 1)^2 / a
    double* fac;     // [Tl][W] (D - 1) log zz
    double* lu;      // [Tl][W] log of the accept uniform
};

// The draws of the second half-step once more, one 32-byte record per moving walker in the order k_split1_pt consumes
// them: workgroup (= column block) b reads records [64 b, 64 b + 64), record m = t * (cb / 2) + (label rank - cb / 2)
// belongs to the m-th moving walker of the block (block_rank) - one coalesced 2 KiB load that depends on nothing.
struct __attribute__((aligned(16))) DrawRec {
    double zz, fac, lu;
    int32_t cw, own;
};

// {log-likelihood, log-prior, pool row} of one walker as ONE 32-byte record (the two-launch iteration of hens_step).
// k_split1_pt finds its 128 walkers through a random column map: in the by-field arrays every field of every walker
// costs a cache line of its own, and the phase that gathers them is bound by the number of lines, not by bytes.
// `acc` is the accept counter of the SLOT the record sits in (move.py:404-421: accepted[t, w]); it does not travel with the
// walker through the cascade.  In record mode the stepping kernels add to it in the record they hold anyway - one
// atomicAdd per accepted proposal on a counter array of its own cost 0.7 us per iteration at config 2 (round 3).
struct __attribute__((aligned(32))) WalkerRec {
    double L, P;
    int32_t loc;
    uint32_t acc;
    int32_t slot;      // column-ordered records: the slot (walker index within the rung) the record belongs to - like `acc` a property
                       // of the place, written once by k_pack_cols and carried along: the slot's next column is ONE inverse
                       // permutation away (k_split1_pt), not a permutation and an inverse
    int32_t pad2;
};
__device__ __forceinline__ WalkerRec make_wrec(double L, double P, int32_t loc, uint32_t acc, int32_t slot = 0) { return WalkerRec{L, P, loc, acc, slot, 0}; }

__device__ __forceinline__ DrawRec draw_values(int own, int cw, double uz, double ua, double a, int D) {
    double zz = (a - 1.0) * uz + 1.0;              // stretch.py:129-132 (mul, add, square, divide)
    zz = zz * zz / a;
    return DrawRec{zz, ((double)D - 1.0) * log(zz) /* stretch.py:223 */, log(ua) /* red_blue.py:294 */, cw, own};
}
// the stretch factor alone - what the proposal needs; (D - 1) log zz and log u are the accept test's (two FP64 logarithms: ~2000
// cycles of dependent ALU that need not sit in front of the first barrier).  Same expressions as draw_values.
__device__ __forceinline__ double draw_zz(double uz, double a) {
    double zz = (a - 1.0) * uz + 1.0;              // stretch.py:129-132 (mul, add, square, divide)
    return zz * zz / a;
}
__device__ __forceinline__ void store_draw(const Draws& d, size_t idx, const DrawRec& r) {
    d.own[idx] = r.own;
    d.cw[idx] = r.cw;
    d.zz[idx] = r.zz;
    d.fac[idx] = r.fac;
    d.lu[idx] = r.lu;
}
__device__ __forceinline__ void make_draw(const Draws& d, size_t idx, int own, int cw, double uz, double ua,
                                          double a, int D) {
    store_draw(d, idx, draw_values(own, cw, uz, ua, a, D));
}

// value of x in lane l, l wave-uniform (v_readlane: a few cycles; __shfl with a runtime lane is a ds_bpermute, ~100)
__device__ __forceinline__ double readlane_f64(double x, int l) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), l);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(x), l);
    return __hiloint2double(hi, lo);
}

// ---------------------------------------------------------------------------------------------
// Ladder adaptation (tempering.py:563-596) from the per-workgroup swap counts of the cascade.
// ---------------------------------------------------------------------------------------------
struct AdaptArgs {
    uint32_t* swap_part;        // [nblocks][T-1] per-workgroup swap counts of the last cascade
    int32_t zero_after;         // the rows are accumulated with atomics (k_split1_pt): the (single) reader clears them
    uint32_t* zero_rows;        // rows to clear besides (the OTHER accumulation buffer: nobody reads or writes it during
                                // this launch), nblocks * (T-1) words, or nullptr
    const double* betas_in;     // [T]
    double* betas_out;          // [T] (may alias betas_in when run as its own kernel)
    double* swaps_last;         // [T-1]
    double* swaps_total;        // [T-1]
    double lag, nu;
    double kappa;               // (lag / (time + lag)) / nu, tempering.py:571-572, formed by the host (the same two IEEE divisions:
                                // wave-uniform, and two fewer FP64 divides on the folded adaptation's dependent chain)
    int64_t time;               // adaptation steps taken so far (tempering.py:596)
    int32_t T, W, nblocks, moving;   // moving: adaptive and not past stop_adaptation (tempering.py:591)
    int32_t row_groups;              // rows a wave sums straight out of memory: 8 per group of 64 / row_groups lanes, the groups
                                     // added with lane exchanges (1: one lane per pair; k_split1_pt accumulates into 8 *
                                     // row_groups rows, see acc_row_groups)
    // ladder pipeline (k_adapt only): the counts come from every rank - wait for their flags first
    const unsigned* wait_flags;      // my mailbox's PF_CNT0.. words, or nullptr
    unsigned* wait_err;
    long long wait_budget;
    uint32_t wait_target;
    int32_t wait_n;
};

// ---------------------------------------------------------------------------------------------
// Stretch half-step: propose + box prior + likelihood + tempered MH test + update, fused.
// ---------------------------------------------------------------------------------------------
struct StretchArgs {
    double* pool;
    int32_t* loc;
    double* L;
    double* P;
    WalkerRec* wrec;           // k_stretch_fast inside hens_step's record mode (else nullptr): {L, P} live in the walker records
    int32_t inplace;           // k_stretch_fast: rows are updated IN PLACE and `loc` is read-only - an accepted proposal
                               // overwrites the walker's current row (nobody reads it during the launch: complements come
                               // from the other set), a rejected one writes nothing.  0: the two-home copying scheme
    const double* betas;       // [T] or nullptr when not tempered
    Draws dr;
    uint32_t* accepted;        // [Tl][W] cumulative accept counts
    uint8_t* keep_out;         // [Tl][Ns] or nullptr
    const double* lo;
    const double* hi;
    const double* period;      // [D] periods of the periodic parameters (0: not periodic), or nullptr: none
    const double* mu;
    const double* prec;
    const double* prec_sym;    // packed symmetric rows for the fast kernel (see sym_quad), dense only
    unsigned* flags;
    unsigned long long* trace; // debug: per-workgroup phase timestamps (s_memtime), or nullptr
    double logp_in, fill, rosen_a, rosen_b;
    int32_t Tl, W, D, split, N0, rung_begin, home_off, tempered, RS;
    int32_t ad_on;             // fold the ladder adaptation of the previous cascade into this launch: 1 = every workgroup
                               // recomputes it; 2 = workgroup (0,0) computes it and publishes every rung's beta in ad_ring
    double* ad_ring;           // [4][T] (mode 2) slot ad_serial & 3 receives the new ladder; -1 = not there yet
    uint32_t ad_serial;
    int64_t guest_delta;       // see row_off (0 when there is no pipeline)
    const double* mh_step;     // MODE_MH: [Tl][W][D] proposal steps, or nullptr: isotropic / axis-aligned steps drawn in place
    const double* mh_scale;    //   (mh_kind MH_ISO: [1], MH_DIAG: [D] standard deviations; Philox keys mh_iter, mh_seed)
    uint64_t mh_iter, mh_seed;
    int32_t mh_kind;
    // ladder pipeline: the hottest resident rung's (L, P) after this move go straight to the hot neighbour
    double* pub_lp;            // neighbour's lp_dn for this sweep: [2][W], or nullptr
    unsigned* pub_flag;        // neighbour's PF_LDN, raised by the last workgroup of the iteration's last launch
    unsigned* pub_ticket;
    uint32_t pub_target, pub_value;
    int32_t pub_final;
    // ladder pipeline, adaptation_delay = 1: the adapting workgroup also reduces the per-workgroup swap counts of the
    // sweep that just ended (rows of k_pipe_walk) and stores the sums into every rank's mailbox + raises their flags
    const uint32_t* cp_rows;   // [cp_nblocks][cp_np]
    char* const* cp_boxes;     // [cp_nranks]
    uint32_t cp_sweep;
    int32_t cnt_push, cp_nblocks, cp_np, cp_nranks, cp_rank, cp_T;
    // (cnt_push 2 - fused pipeline iteration on the reference's adaptation schedule: the counts are due in THIS launch, so wave 1
    //  of workgroup (0,0) sums and publishes them before anything else, in front of the wait for every rank's counts.
    //  cp_zero: the rows were accumulated with atomics by k_split1_pt<PIPE> - the (sole) reader clears them)
    int32_t cp_zero;
    unsigned* rt_flag;         // fused pipeline iteration: the cold neighbour's PF_ROWS_TOP, raised at the head of this launch - the
    uint32_t rt_value;         //   kernel boundary says that ALL pushes / pulls of the previous sweep's bottom phase are complete
    int32_t sys_rung;          // local rung whose rows a peer will read (written through to memory, system scope), or -1
    // fused pipeline iteration (in-place rows on a pipeline rank): a walker that arrived through the pipeline sits in a guest
    // row (loc < 0) until its next half-step, which writes its row - the proposal or the old one - into the pool row the walker
    // that left in exchange has vacated: ghome[~loc]; sys_all: every row store is system scope (a row is no longer rewritten
    // every iteration, and any of them may end up in the rung a peer pulls from)
    const int32_t* ghome;      // [2][2][W] home row of the guest at (sweep parity, side, column), or nullptr
    int32_t sys_all;
    long long* pub_meta;       // neighbour's meta[par]: receives the pool row of (hottest rung, slot 0) after this move
    // ladder pipeline: before touching the state, wait until the mailbox flags selected by wmask reach wtarget
    const unsigned* wflags;
    unsigned long long wmask;
    long long wbudget;
    uint32_t wtarget, wtarget_cnt;   // rows flags / swap-count flags (PF_CNT0..)
    unsigned long long* wstats;      // debug wait statistics or nullptr
    // hens_step's two-launch iteration on block-balanced labels: no planned draws - the launch computes them in registers
    // from the iteration's round keys (stretch_draws_at; `dr` is not read)
    const uint32_t* ikeys;           // [T][8] round keys of this iteration's column maps (k_plan_keys), or nullptr
    uint64_t iseed, iiter;
    double ia;                       // stretch scale a
    int32_t idx_bits, hb_shift, ndim_active;
    // Column-ordered records (round 3, hens_step's two launches on one GPU): wrec / loc hold every rung in the order of THIS
    // iteration's cascade columns - record c of rung t belongs to the walker column c meets - so the walker at place p of a
    // half is record place_column(half, p): a coalesced load that waits for no round key and no permutation, and a complement
    // is a lookup by column in the rung's compact row table (`loc`, 4 W bytes per rung: L2-resident).
    // k_split1_pt<COL> writes the next buffers in the NEXT iteration's column order (scattered stores at its tail instead of
    // scattered loads at both launches' heads).
    int32_t col;
    int32_t inject_c64;        // dev hook (HENS_PIPE_INJECT_CYCLES, pipeline ranks): the swap-count flags of the previous sweep count as raised
                               // only that many x 64 shader cycles after the adapting workgroup's start - a neighbour whose counts
                               // arrive that late; 0: off.  Measures the slack the first launch absorbs (tools/pipe_slack.sh).
                               // -1 (HENS_PIPE_FORCE_LATE=1, any build): the adapting wave always comes back for the counts (tests)
    // parity API with nsplits > 2 (red_blue.py:41-47,148): the moving set's position range, given explicitly (0: the two-half
    // rule from N0 / split); `split` is then 0 for the first set, 1 for the last (every complement already sits in its home
    // row) and 2 for the ones between
    int32_t ns_x, soff_x;
    int32_t norel;             // this launch's packet carries no release fence (hens_aql.h: norel_next): the records are written through and
                               // every wave ends behind its stores (wt_store, launch_end_wait); 0: plain stores, the fence writes them back
    int32_t xcd_shift;         // > 0: log2(tiles per rung) + 1 - workgroups are renumbered so that an XCD (linear id mod 8) works on whole rungs
    int32_t tiles_per_wg;      // k_stretch2 (hens_tile2.h): tiles a persistent workgroup walks (grid.x = tiles per rung / tiles_per_wg)
    AdaptArgs ad;
};

// One workgroup = TILE (64) walkers of one rung, NW wavefronts.  Generic row width (runtime D).
//   phase A  lane-per-walker : indices, draws                              (wave 0)
//   phase B  lanes-over-d    : coalesced row gathers, q = c-(c-s)zz, box test by ballot -> LDS
//   phase C  lane-per-walker : quadratic form, rows of the precision matrix split over the waves
//   phase D  lane-per-walker : tempered MH test, L/P/loc/accept counters
//   phase E  lanes-over-d    : coalesced write of the new row (q if kept, old row otherwise)
template <int LIKE, int MODE>
__global__ __launch_bounds__(256) void k_stretch(const StretchArgs A) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    constexpr bool EVAL = MODE == MODE_EVAL, MH = MODE == MODE_MH;
    constexpr int NW = 4, NT = 256;
    const int D = A.D, RS = A.RS;
    double* qtile = reinterpret_cast<double*>(smem_raw);                 // [TILE][RS]
    double* s_zz = qtile + TILE * RS;                                    // [TILE]
    double* s_part = s_zz + TILE;                                        // [NW][TILE]
    int32_t* s_rs = reinterpret_cast<int32_t*>(s_part + NW * TILE);      // [TILE] own row
    int32_t* s_rc = s_rs + TILE;                                         // [TILE] complement row
    int32_t* s_dst = s_rc + TILE;                                        // [TILE] destination row
    int32_t* s_flag = s_dst + TILE;                                      // bit0 inbox, bit1 keep, bit2 valid

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tl = blockIdx.y;
    const int W = A.W;
    const int Ns = (EVAL || MH) ? W : (A.ns_x ? A.ns_x : (A.split == 0 ? A.N0 : W - A.N0));
    const int s_off = (EVAL || MH) ? 0 : (A.ns_x ? A.soff_x : (A.split == 0 ? 0 : A.N0));
    const int k0 = blockIdx.x * TILE;

    double factors = 0.0, lu = 0.0, Lold = 0.0, Pold = 0.0;
    int own = 0;
    bool valid = false;
    if (wv == 0) {
        const int k = k0 + lane;
        valid = k < Ns;
        double zz = 1.0;
        int rs = 0, rc = 0;
        if (valid) {
            if (EVAL) {
                own = k;
                rs = A.loc[tl * W + own];
                rc = rs;
            } else if (MH) {                     // every walker proposes; no partner, no Hastings factor
                own = k;
                rs = A.loc[tl * W + own];
                rc = rs;
                lu = A.dr.lu[(size_t)tl * W + own];
                Lold = A.L[tl * W + own];
                Pold = A.P[tl * W + own];
            } else {
                const size_t di = (size_t)tl * W + s_off + k;
                own = A.dr.own[di];
                const int cw = A.dr.cw[di];
                zz = A.dr.zz[di];
                factors = A.dr.fac[di];
                lu = A.dr.lu[di];
                rs = A.loc[tl * W + own];
                rc = A.split == 1 ? A.home_off + tl * W + cw : A.loc[tl * W + cw];
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

    // ---- phase B: VEC doubles per lane; LPR lanes per row (power of two <= 64); RPP rows per pass
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
        const bool rvalid = (r < TILE) && (s_flag[r < TILE ? r : 0] & 4) != 0;   // RPP may exceed TILE for tiny D
        bool ok = true, finite = true;
        if (rvalid) {
            const double zz = s_zz[r];
            const double* ps = pool_r + row_off(s_rs[r], D, A.guest_delta);
            const double* pc = MH ? A.mh_step + ((size_t)tl * W + k0 + r) * D : pool_r + row_off(s_rc[r], D, A.guest_delta);
            for (int ch = jl; ch < chunks; ch += LPR) {
                const int e = ch * VEC;
                if (VEC == 2) {
                    const double2 sv = *reinterpret_cast<const double2*>(ps + e);
                    double2 qv;
                    if (EVAL) {
                        qv = sv;
                    } else {
                        const double2 cv = *reinterpret_cast<const double2*>(pc + e);
                        if (MH) {
                            qv.x = sv.x + cv.x;               // gaussian.py:166-167
                            qv.y = sv.y + cv.y;
                        } else if (A.period) {                // stretch.py:136-145 with periodic parameters
                            qv.x = cv.x - periodic_diff(sv.x, cv.x, A.period[e]) * zz;
                            qv.y = cv.y - periodic_diff(sv.y, cv.y, A.period[e + 1]) * zz;
                        } else {
                            qv.x = cv.x - (cv.x - sv.x) * zz; // stretch.py:143,145
                            qv.y = cv.y - (cv.y - sv.y) * zz;
                        }
                        if (A.period) {                       // stretch.py:149-154, gaussian.py:110-115
                            qv.x = periodic_wrap(qv.x, A.period[e]);
                            qv.y = periodic_wrap(qv.y, A.period[e + 1]);
                        }
                    }
                    const double2 lov = *reinterpret_cast<const double2*>(A.lo + e);
                    const double2 hiv = *reinterpret_cast<const double2*>(A.hi + e);
                    ok = ok && (qv.x >= lov.x) && (qv.x <= hiv.x) && (qv.y >= lov.y) && (qv.y <= hiv.y);
                    finite = finite && (fabs(qv.x) < INFINITY) && (fabs(qv.y) < INFINITY);
                    *reinterpret_cast<double2*>(qtile + r * RS + e) = qv;
                } else {
                    const double sv = ps[e];
                    double qv;
                    if (EVAL) {
                        qv = sv;
                    } else {
                        const double cv = pc[e];
                        if (MH) qv = sv + cv;
                        else if (A.period) qv = cv - periodic_diff(sv, cv, A.period[e]) * zz;
                        else qv = cv - (cv - sv) * zz;
                        if (A.period) qv = periodic_wrap(qv, A.period[e]);
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

    // ---- phase C
    {
        const bool inbox = (s_flag[lane] & 1) != 0;
        double part = 0.0;
        typedef const __attribute__((address_space(4))) double* cptr_t;   // read-only for the launch: scalar loads
        const cptr_t mu = (cptr_t)(uintptr_t)A.mu;
        const cptr_t prec = (cptr_t)(uintptr_t)A.prec;
        const double* qrow = qtile + lane * RS;
        if (LIKE == LIKE_HOST) {
            part = 0.0;                // the caller evaluates the likelihood (hens_propose_split / hens_accept_split)
        } else if (LIKE == LIKE_ROSEN) {
            if (wv == 0 && inbox) {
                double acc = 0.0;
                for (int i = 0; i + 1 < D; ++i) {
                    const double x0 = qrow[i], x1 = qrow[i + 1];
                    const double t1 = x1 - x0 * x0, t2 = A.rosen_a - x0;
                    acc += A.rosen_b * (t1 * t1) + t2 * t2;
                }
                part = 2.0 * acc;      // phase D multiplies by -0.5
            }
        } else if (inbox) {
            const int RB = (D + NW - 1) / NW;
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
        s_part[wv * TILE + lane] = part;
    }
    __syncthreads();

    // ---- phase D
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
        if (EVAL) {
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
    if (EVAL) return;
    __syncthreads();

    // ---- phase E
    double* __restrict__ pool_w = A.pool;
    for (int r0 = 0; r0 < TILE; r0 += RPP) {
        const int r = r0 + rsub;
        if (r >= TILE) continue;
        const int fl = s_flag[r];
        if (!(fl & 4)) continue;
        const bool keep = (fl & 2) != 0;
        double* pd = pool_w + (size_t)s_dst[r] * D;
        const double* ps = pool_r + row_off(s_rs[r], D, A.guest_delta);
        for (int ch = jl; ch < chunks; ch += LPR) {
            const int e = ch * VEC;
            if (VEC == 2) {
                const double2 v = keep ? *reinterpret_cast<const double2*>(qtile + r * RS + e)
                                       : *reinterpret_cast<const double2*>(ps + e);
                if (tl == A.sys_rung) {                 // the hot neighbour pulls rows out of this rung (ladder pipeline)
                    sys_store(pd + e, v.x);
                    sys_store(pd + e + 1, v.y);
                } else {
                    *reinterpret_cast<double2*>(pd + e) = v;
                }
            } else {
                const double v = keep ? qtile[r * RS + e] : ps[e];
                if (tl == A.sys_rung) sys_store(pd + e, v);
                else pd[e] = v;
            }
        }
    }
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt, i.e. it waits for
// every outstanding global STORE (CDNA4 counts stores in vmcnt); the stretch kernel deliberately leaves
// row stores in flight across its phases.  Global loads are still waited for where their values are used.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// DEV BUILDS ONLY (tools/devbuild.sh -DHENS_CUT_S=n / -DHENS_CUT_F=n; compiled out of the product): the first / second launch returns after phase n - cumulative
// phase timings without trace stamps (results are wrong; run with HENS_DEBUG_NOFLIP=1 so the state stays addressable)
#ifndef HENS_CUT_S
#define HENS_CUT_S 0
#endif
#ifndef HENS_CUT_F
#define HENS_CUT_F 0
#endif

// 16-byte row store, variant selected at build time for the experiment:
//   HENS_ROWSTORE 0 plain (write-back L2), 1 sc1 write-through (asm), 2 nontemporal
#ifndef HENS_ROWSTORE
#define HENS_ROWSTORE 1
#endif
// The stepping launches of one GPU carry no release fence (hens_aql.h: norel_next; hens.hip: norel_ok) BECAUSE a row leaves as a
// write-through store: with HENS_ROWSTORE 0 or 2 (A/B libraries) rows would sit dirty in one XCD's L2 and the next launch would
// read stale data on another - norel_ok() asks this constant and keeps the fence then.
constexpr bool ROWSTORE_WRITE_THROUGH = HENS_ROWSTORE == 1;
typedef double dvec2 __attribute__((ext_vector_type(2)));
// system scope (sc0 sc1): written through to memory - for rows a peer GPU reads (an agent-scope store may sit
// dirty in this GPU's L2, which a read arriving over xGMI does not probe)
__device__ __forceinline__ void store_row16_sys(double* p, double2 v) {
    const double __attribute__((ext_vector_type(2))) t = {v.x, v.y};
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(t) : "memory");
}
__device__ __forceinline__ void store_row16(double* p, double2 v) {
#if HENS_ROWSTORE == 1
    const dvec2 t = {v.x, v.y};
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(t) : "memory");
#elif HENS_ROWSTORE == 2
    const dvec2 t = {v.x, v.y};
    __builtin_nontemporal_store(t, reinterpret_cast<dvec2*>(p));
#else
    *reinterpret_cast<double2*>(p) = v;
#endif
}
// a walker record, written through: {L, P} and {loc, acc, slot, -} as two 16-byte stores
__device__ __forceinline__ void wt_store_rec(WalkerRec* p, const WalkerRec& r) {
    store_row16(&p->L, double2{r.L, r.P});
    const double lo = __hiloint2double((int)r.acc, r.loc), hi = __hiloint2double(0, r.slot);
    store_row16(reinterpret_cast<double*>(&p->loc), double2{lo, hi});
}

// Symmetric quadratic form, the share of wave WI.  psym holds, for every row pair p = 0..D/2-1,
// D + 2 doubles: row p from its diagonal rightwards [A_pp, A_p,p+1 + A_p+1,p, ...] (D - p entries)
// followed by row D-1-p likewise (p + 1 entries), then one pad.  Wave WI owns the pairs
// WI*PPW .. WI*PPW + PPW - 1.
template <int DT, int NW, int WI>
__device__ __forceinline__ double sym_quad(const double (&q)[DT], const __attribute__((address_space(4))) double* psym) {
    constexpr int PPW = (DT / 2) / NW;                 // row pairs per wave
    static_assert(PPW >= 1 && (DT / 2) % NW == 0, "row pairs must divide over the waves");
    double part = 0.0;
    if (WI >= NW) return part;
#pragma unroll
    for (int pp = 0; pp < PPW; ++pp) {
        constexpr int STRIDE = DT + 2;
        const int p = (WI < NW ? WI : 0) * PPW + pp;
        const __attribute__((address_space(4))) double* c = psym + p * STRIDE;
        double y = 0.0;
#pragma unroll
        for (int t = 0; t < DT; ++t)
            if (t < DT - p) y = fma(c[t], q[p + t < DT ? p + t : 0], y);
        part = fma(q[p], y, part);
        const int i2 = DT - 1 - p;
        double y2 = 0.0;
#pragma unroll
        for (int t = 0; t < DT; ++t)
            if (t <= p) y2 = fma(c[DT - p + t], q[i2 + t < DT ? i2 + t : 0], y2);
        part = fma(q[i2], y2, part);
    }
    return part;
}

// Phase C of the stretch kernels: this wave's share of -2 log-likelihood of the proposal in row `lane` of the LDS tile
// (lane per walker; the rows / blocks of the precision matrix are dealt to the NW waves, see sym_quad).  The
// coefficients come through scalar loads (SGPR operands).  Returns the partial sum; the caller adds the NW parts.
// For the Gaussian likelihoods the tile holds the CENTRED proposal q - mu (phase B subtracts once per element while it
// has the element in a register; every wave subtracting for itself was a quarter of this phase's FP64 issue slots).
// (Rows up to 32 doubles: wider rows need the registers that would hold the proposal for phase E - 135 VGPRs at
// D = 64 instead of 119, one workgroup per CU instead of two - and keep the uncentred tile.)
constexpr bool like_centred(int LIKE, int DT) { return (LIKE == LIKE_DENSE || LIKE == LIKE_DIAG) && DT <= 32; }
template <int DT, int LIKE, int NW, bool CEN>
__device__ __forceinline__ double like_partial(const double* qtile, int lane, int wv, bool inbox, const double* mu_p,
                                               const double* prec_p, const double* prec_sym_p, double rosen_a, double rosen_b) {
    constexpr int D = DT, RS = DT + 2;
    double part = 0.0;
    typedef const __attribute__((address_space(4))) double* cptr_t;   // read-only for the launch: SGPR scalar loads
    const cptr_t mu = (cptr_t)(uintptr_t)mu_p;
    const cptr_t prec = (cptr_t)(uintptr_t)prec_p;
    const double* qrow = qtile + lane * RS;
    if (LIKE == LIKE_ROSEN) {
        // the D - 1 coupled terms dealt to the NW waves in contiguous runs (one wave alone took ~7600 cycles at D = 128
        // while the others idled); the caller adds the parts in wave order
        constexpr int PER = (DT - 1 + NW - 1) / NW;
        if (inbox && wv < NW) {
            double acc = 0.0;
            const int i0 = wv * PER, i1 = (i0 + PER < D - 1) ? i0 + PER : D - 1;
            for (int i = i0; i < i1; ++i) {
                const double x0 = qrow[i], x1 = qrow[i + 1];
                const double t1 = x1 - x0 * x0, t2 = rosen_a - x0;
                acc += rosen_b * (t1 * t1) + t2 * t2;
            }
            part = 2.0 * acc;
        }
    } else if (inbox) {
        constexpr int RB = (DT + NW - 1) / NW;
        const int i0 = wv * RB;
        if (LIKE == LIKE_DENSE && (DT == 64 || DT == 128) && NW == 8) {
            // D = 64 / 128: a lane holding all the centred coordinates needs 2 D VGPRs for them alone (D = 64: 165 in
            // total, one workgroup per CU).  Blocked instead into four H = D / 4 blocks: 4 symmetric diagonal blocks
            // (sym_quad<H>) and 6 cross blocks (H x H, A_ik + A_ki).  Waves 0-5 take one cross block each, waves 6 and 7 two
            // diagonal blocks each (D = 64: 272 / 304 FMAs per lane, D = 128: 1024 / 1056), one H-vector in registers at
            // a time - at D = 64 that is 32 VGPRs instead of the 64 of a two-block form.
            constexpr int H = DT / 4, SB = (H / 2) * (H + 2);
            const cptr_t psym = (cptr_t)(uintptr_t)prec_sym_p;
            double qh[H];
            if (wv < 6) {
                const int bi = wv < 3 ? 0 : (wv < 5 ? 1 : 2);                  // (0,1) (0,2) (0,3) (1,2) (1,3) (2,3)
                const int bk = wv < 3 ? wv + 1 : (wv < 5 ? wv - 1 : 3);
#pragma unroll
                for (int k = 0; k < H; k += 2) {
                    const double2 v = *reinterpret_cast<const double2*>(qrow + bk * H + k);
                    qh[k] = v.x - mu[bk * H + k];
                    qh[k + 1] = v.y - mu[bk * H + k + 1];
                }
                const cptr_t cx = psym + 4 * SB + wv * H * H;
#pragma unroll 4
                for (int r = 0; r < H; ++r) {
                    double y = 0.0;
#pragma unroll
                    for (int k = 0; k < H; ++k) y = fma(cx[r * H + k], qh[k], y);
                    part = fma(qrow[bi * H + r] - mu[bi * H + r], y, part);
                }
            } else {
#pragma unroll
                for (int b2 = 0; b2 < 2; ++b2) {
                    const int b = (wv - 6) * 2 + b2;
#pragma unroll
                    for (int k = 0; k < H; k += 2) {
                        const double2 v = *reinterpret_cast<const double2*>(qrow + b * H + k);
                        qh[k] = v.x - mu[b * H + k];
                        qh[k + 1] = v.y - mu[b * H + k + 1];
                    }
                    part += sym_quad<H, 1, 0>(qh, psym + b * SB);
                }
            }
        } else if (LIKE == LIKE_DENSE) {
            double qreg[DT];
#pragma unroll
            for (int k = 0; k < DT; k += 2) {
                const double2 v = *reinterpret_cast<const double2*>(qrow + k);
                qreg[k] = CEN ? v.x : v.x - mu[k];
                qreg[k + 1] = CEN ? v.y : v.y - mu[k + 1];
            }
            // q'^T A q' = sum_i q'_i (A_ii q'_i + sum_{k>i} (A_ik + A_ki) q'_k): half the FP64 FMAs of
            // the full product (phase C is FP64-rate-bound: vector fp64 = 78.6 TF/s).  Rows are dealt
            // to the waves in pairs (p, D-1-p) of equal total length; the wave index becomes a
            // compile-time constant through the switch so every register index is static.
            const cptr_t psym = (cptr_t)(uintptr_t)prec_sym_p;
            switch (wv) {
#define HENS_SYM_CASE(WI) case WI: part = sym_quad<DT, NW, WI>(qreg, psym); break;
                HENS_SYM_CASE(0) HENS_SYM_CASE(1) HENS_SYM_CASE(2) HENS_SYM_CASE(3)
                HENS_SYM_CASE(4) HENS_SYM_CASE(5) HENS_SYM_CASE(6) HENS_SYM_CASE(7)
#undef HENS_SYM_CASE
                default: break;
            }
        } else {
            for (int ii = 0; ii < RB; ++ii) {
                const int i = i0 + ii;
                if (i < DT) {
                    const double di = CEN ? qrow[i] : qrow[i] - mu[i];
                    part = fma(di * prec[i], di, part);
                }
            }
        }
    }
    return part;
}

// ---- dense Gaussian at D = 64, round 4: the quadratic form on the FP64 MATRIX pipe ------------------------------------------------
// Measured on gfx950 (tools/probe/mfma_f64_rate.hip): a wave issues v_fma_f64 once per 8.5 cycles (7.5 multiply-adds per clock and
// SIMD), v_mfma_f64_16x16x4_f64 once per 64 cycles (16 per clock and SIMD).  Phase C at D = 64 is bound by exactly that issue rate:
// cut builds at 8 x 16384 x 64 give 8.1 us of a 27.0 us first launch (7.2 of 24.7 in the second) for the blocked symmetric VALU form
// - 8 waves x ~290 FMAs per tile, 19 700 SIMD-cycles, the launch's 1 024 tiles on 1 024 SIMDs.  (At D = 32 the same phase is bound
// by LDS reads and the matrix pipe bought nothing: tools/probe/mfma_like_d32_*.)  With the centred proposal q cut into four blocks
// of 16 coordinates,
//     q^T A q = sum_I q_I . y_I,     y_I = sum_{J >= I} M_IJ q_J,     M_II = A_II,  M_IJ = A_IJ + A_JI^T  (J > I)
// is 10 block products of 16 x 16, each 4 MFMA steps of 16 x 16 x 4 per 16 walkers: A operand M_IJ[r][4 s + k] (the same for every
// workgroup: prec_sym's last part, mf[step][lane] = M[lane % 16][4 s + lane / 16], requested before the barrier), B operand
// q[16 J + 4 s + k][walker] out of the LDS tile (the tile holds q itself - phase E stores accepted rows from it; centring it would
// cost 16 registers for the proposal across phase C, one workgroup per CU - so mu is subtracted on the way, out of a 512-byte LDS
// copy of its own: MF_LDS_EXTRA behind each kernel's arrays).  The 40 (I, J, s) steps are dealt to the eight waves five each, in order - a wave
// meets at most two row blocks I - and every wave runs its steps for all four 16-walker blocks (20 MFMAs: 1 280 cycles of its SIMD's
// matrix pipe, two waves per SIMD; the VALU form took 4 900 per SIMD), then dots its partial y with q_I: the result tile holds rows
// g, g + 4, g + 8, g + 12 of walker l % 16 in lane l = 16 g + l % 16 (tools/probe/mfma_f64_layout.cpp), so 4 FMAs per row block
// and one sum over the four 16-lane rows (gfx950's row swaps, VALU only).  Eight partial sums per walker, added in wave order in
// phase D as before.  Same value as the VALU form to ~1e-15 relative (another summation order; the oracle comparison has 1e-13).
// EVERY dense D = 64 path goes through here (k_stretch_fast in all its modes and k_split1_pt, single GPU and pipeline rank), so they
// stay bit-identical to one another.
constexpr int MF64_OFF = 4 * ((16 / 2) * (16 + 2)) + 6 * 16 * 16;        // behind the blocked sym_quad form in prec_sym (D = 64)
constexpr int MF64_STEPS = 40;
constexpr int MF64_END = MF64_OFF + MF64_STEPS * 64;
// D = 128: eight blocks of 16 coordinates, 36 block products, 144 steps - 18 per wave (like_tile_mf128)
constexpr int MF128_OFF = 4 * ((32 / 2) * (32 + 2)) + 6 * 32 * 32;       // behind the blocked sym_quad form in prec_sym (D = 128)
constexpr int MF128_STEPS = 144;
constexpr int MF128_END = MF128_OFF + MF128_STEPS * 64;
constexpr int mf128_base(int I) { return 4 * (I * 8 - I * (I - 1) / 2); }                 // steps in front of row block I
constexpr int mf128_I(int c) { int I = 0; while (I < 7 && c >= mf128_base(I + 1)) ++I; return I; }
constexpr int mf128_J(int c) { return mf128_I(c) + (c - mf128_base(mf128_I(c))) / 4; }
constexpr int mf128_S(int c) { return (c - mf128_base(mf128_I(c))) % 4; }
// mu in LDS, dense likelihood only (D = 128 on a pipeline rank: with it for every kind the diagonal likelihood's 127-VGPR launch would
// lose its second workgroup per CU to 512 bytes of LDS)
// D = 32 (round 5): two blocks of 16 coordinates, 3 block products = 12 steps (I, J >= I, s) behind the pair layout in prec_sym
// (like_tile_mf32).  HENS_NO_MF32 builds the scalar-operand VALU form (sym_quad) instead.
constexpr int MF32_OFF = 640;                                            // (17 x 34 = 578 doubles of pair layout in front, rounded up)
constexpr int MF32_STEPS = 12;
constexpr int MF32_END = MF32_OFF + MF32_STEPS * 64;
constexpr int mf32_I(int c) { return c < 8 ? 0 : 1; }
constexpr int mf32_J(int c) { return c < 4 ? 0 : 1; }
constexpr int mf32_S(int c) { return c & 3; }
#ifdef HENS_NO_MF32
constexpr bool MF32_ON = false;
#else
constexpr bool MF32_ON = true;
#endif
__host__ __device__ constexpr size_t mf_lds_extra(int D, int like) {
    return like != LIKE_DENSE ? 0 : (D == 64 ? 64 * 8 : (D == 128 ? 128 * 8 : (D == 32 && MF32_ON ? 32 * 8 : 0)));
}
typedef double d4_t __attribute__((ext_vector_type(4)));
struct MfRegs { double m[6]; };
template <int DT, int LIKE, int NW>
constexpr bool like_mf() { return LIKE == LIKE_DENSE && (DT == 64 || DT == 128 || (DT == 32 && MF32_ON)) && NW == 8; }
// partial sums per walker phase C leaves in s_part[part][walker] (phase D adds them in order)
template <int DT, int LIKE, int NW>
constexpr int like_nparts() { return (LIKE == LIKE_DENSE && DT == 32 && MF32_ON && NW == 8) ? 2 : NW; }
// step c = 0..39 in the order (I, J >= I, s)
constexpr int mf64_I(int c) { return c < 16 ? 0 : (c < 28 ? 1 : (c < 36 ? 2 : 3)); }
constexpr int mf64_base(int I) { return I == 0 ? 0 : (I == 1 ? 16 : (I == 2 ? 28 : 36)); }
constexpr int mf64_J(int c) { return mf64_I(c) + (c - mf64_base(mf64_I(c))) / 4; }
constexpr int mf64_S(int c) { return (c - mf64_base(mf64_I(c))) % 4; }
// the matrix operand of the wave's five steps, requested before the barrier in front of phase C (a no-op elsewhere)
template <int DT, int LIKE, int NW>
__device__ __forceinline__ MfRegs like_prefetch(int lane, int wv, const double* prec_sym_p) {
    MfRegs r;
#pragma unroll
    for (int t = 0; t < 6; ++t) r.m[t] = 0.0;
    if constexpr (like_mf<DT, LIKE, NW>() && DT == 64) {          // (D = 128: 18 operands per wave, requested where they are used)
#pragma unroll
        for (int t = 0; t < 5; ++t) r.m[t] = prec_sym_p[MF64_OFF + (5 * wv + t) * 64 + lane];
    }
    if constexpr (like_mf<DT, LIKE, NW>() && DT == 32) {          // (waves 0-3: steps 0-5, waves 4-7: steps 6-11 - see like_tile_mf32)
#pragma unroll
        for (int t = 0; t < 6; ++t) r.m[t] = prec_sym_p[MF32_OFF + (6 * (wv >> 2) + t) * 64 + lane];
    }
    return r;
}
// sum over the four 16-lane rows of a wave, in every lane (lanes l, l ^ 16, l ^ 32, l ^ 48): gfx950's row swaps - VALU only, no LDS
// round trip
__device__ __forceinline__ double sum_rows_f64(double p) {
    unsigned lo = (unsigned)__double2loint(p), hi = (unsigned)__double2hiint(p);
    const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    const double x = __hiloint2double((int)b[0], (int)a[0]) + __hiloint2double((int)b[1], (int)a[1]);
    lo = (unsigned)__double2loint(x); hi = (unsigned)__double2hiint(x);
    const auto c = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    const auto d = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    return __hiloint2double((int)d[0], (int)c[0]) + __hiloint2double((int)d[1], (int)c[1]);
}
// wave WI's share for the tile's 64 walkers: lane l returns walker l's partial sum.  CEN: the tile holds q - mu.
template <int WI, bool CEN>
__device__ __forceinline__ double mf64_wave(const double* qtile, int lane, const double* mu_p, const MfRegs& mf) {
    constexpr int RS = 66, C0 = 5 * WI;
    constexpr int IA = mf64_I(C0), IB = mf64_I(C0 + 4);          // the (at most two) row blocks of this wave's steps
    const int j = lane & 15, g = lane >> 4;
    double mk[5] = {0.0, 0.0, 0.0, 0.0, 0.0}, mra[4] = {0.0, 0.0, 0.0, 0.0}, mrb[4] = {0.0, 0.0, 0.0, 0.0};
    if (!CEN) {
#pragma unroll
        for (int t = 0; t < 5; ++t) mk[t] = mu_p[16 * mf64_J(C0 + t) + 4 * mf64_S(C0 + t) + g];
#pragma unroll
        for (int r = 0; r < 4; ++r) { mra[r] = mu_p[16 * IA + g + 4 * r]; mrb[r] = mu_p[16 * IB + g + 4 * r]; }
    }
    double psel = 0.0;
#pragma unroll
    for (int b = 0; b < 4; ++b) {                                // (one accumulator chain per block: the pipe issues a dependent MFMA
        const double* qw = qtile + (16 * b + j) * RS;            //  as fast as an independent one, and four chains cost 64 VGPRs)
        d4_t accA = d4_t{0.0, 0.0, 0.0, 0.0}, accB = d4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int t = 0; t < 5; ++t) {
            const double bq = qw[16 * mf64_J(C0 + t) + 4 * mf64_S(C0 + t) + g] - mk[t];      // this lane's coordinate of the step
            if (mf64_I(C0 + t) == IA) accA = __builtin_amdgcn_mfma_f64_16x16x4f64(mf.m[t], bq, accA, 0, 0, 0);
            else accB = __builtin_amdgcn_mfma_f64_16x16x4f64(mf.m[t], bq, accB, 0, 0, 0);
        }
        double p = (qw[16 * IA + g] - mra[0]) * accA[0];         // rows g, g + 4, g + 8, g + 12 of walker 16 b + j
        p = fma(qw[16 * IA + g + 4] - mra[1], accA[1], p);
        p = fma(qw[16 * IA + g + 8] - mra[2], accA[2], p);
        p = fma(qw[16 * IA + g + 12] - mra[3], accA[3], p);
        if (IB != IA) {
            p = fma(qw[16 * IB + g] - mrb[0], accB[0], p);
            p = fma(qw[16 * IB + g + 4] - mrb[1], accB[1], p);
            p = fma(qw[16 * IB + g + 8] - mrb[2], accB[2], p);
            p = fma(qw[16 * IB + g + 12] - mrb[3], accB[3], p);
        }
        p = sum_rows_f64(p);
        psel = (g == b) ? p : psel;                              // (lane group g keeps block g's walkers: lane l = walker l)
    }
    return psel;
}
// one 64-walker tile: the eight partial sums of every walker into srow[wave * TILE + walker]
template <bool CEN>
__device__ __forceinline__ void like_tile_mf64(const double* qtile, double* srow, int lane, int wv, const double* mu_p, const MfRegs& mf) {
    double part = 0.0;
    switch (wv) {
#define HENS_MF_CASE(WI) case WI: part = mf64_wave<WI, CEN>(qtile, lane, mu_p, mf); break;
        HENS_MF_CASE(0) HENS_MF_CASE(1) HENS_MF_CASE(2) HENS_MF_CASE(3)
        HENS_MF_CASE(4) HENS_MF_CASE(5) HENS_MF_CASE(6) HENS_MF_CASE(7)
#undef HENS_MF_CASE
        default: break;
    }
    srow[wv * TILE + lane] = part;
}

// D = 32, the same scheme (round 5).  The scalar-operand form was bound by latency, not by issue: 136 coefficients per computing wave
// through ~80 free SGPRs = eight batches of s_load_dwordx16 with a full wait each (SMEM returns out of order: there is no partial
// wait), 2 070 cycles for phase C with both workgroups of a CU resident for the whole launch (tools/trace_fused2.py).  Here: 12 steps
// (I, J >= I, s) x four 16-walker blocks = 48 MFMAs, six per wave - wave w takes walker block w & 3 and the steps' half w >> 2
// ([0, 6): row block 0's products with column blocks 0 and half of 1; [6, 12): the rest of it and row block 1's own) - so a lane
// reads 6 + 4 (+ 4) doubles of the tile where it read 32, and the six matrix operands come in one vector load each, requested in
// front of the barrier.  TWO partial sums per walker (like_nparts): s_part[half][walker].
template <int H, bool CEN>
__device__ __forceinline__ double mf32_half(const double* qw, int g, const double* mu_s, const MfRegs& mf) {
    constexpr int C0 = 6 * H;
    d4_t accA = d4_t{0.0, 0.0, 0.0, 0.0}, accB = d4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int t = 0; t < 6; ++t) {
        const int k = 16 * mf32_J(C0 + t) + 4 * mf32_S(C0 + t) + g;          // this lane's coordinate of the step
        double bq = qw[k];
        if (!CEN) bq -= mu_s[k];
        if (mf32_I(C0 + t) == 0) accA = __builtin_amdgcn_mfma_f64_16x16x4f64(mf.m[t], bq, accA, 0, 0, 0);
        else accB = __builtin_amdgcn_mfma_f64_16x16x4f64(mf.m[t], bq, accB, 0, 0, 0);
    }
    double p = 0.0;                                                           // rows g, g + 4, g + 8, g + 12 of the walker
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        double x = qw[g + 4 * r];
        if (!CEN) x -= mu_s[g + 4 * r];
        p = r == 0 ? x * accA[0] : fma(x, accA[r], p);
    }
    if (H == 1) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            double x = qw[16 + g + 4 * r];
            if (!CEN) x -= mu_s[16 + g + 4 * r];
            p = fma(x, accB[r], p);
        }
    }
    return sum_rows_f64(p);
}
// one 64-walker tile: the two partial sums of every walker into srow[half * TILE + walker]
template <bool CEN>
__device__ __forceinline__ void like_tile_mf32(const double* qtile, double* srow, int lane, int wv, const double* mu_s, const MfRegs& mf) {
    const int j = lane & 15, g = lane >> 4, b = wv & 3;
    const double* qw = qtile + (16 * b + j) * 34;
    const double p = (wv < 4) ? mf32_half<0, CEN>(qw, g, mu_s, mf) : mf32_half<1, CEN>(qw, g, mu_s, mf);
    if (g == 0) srow[(wv >> 2) * TILE + 16 * b + j] = p;
}

// D = 128, the same scheme: 36 block products = 144 steps, 18 per wave in the order (I, J >= I, s); a wave meets up to three row
// blocks (one accumulator chain each per 16-walker block), its matrix operands come straight from prec_sym's last part (L2), mu
// from the LDS copy.  72 MFMAs per wave - 4 608 cycles of its SIMD's matrix pipe - where the blocked VALU form issued ~1 040 FMAs
// per lane (8 900 cycles) and held 32 coordinates in registers (156 - 197 VGPRs: one workgroup per CU).
// one row block I's steps [T0, T1) of the wave for the four 16-walker blocks of the tile: the matrix operand is requested once per
// step and used four times (four independent accumulator chains), then the partial y_I is dotted with q_I into p[block]
template <int C0, int T0, int T1, int I>
__device__ __forceinline__ void mf128_segment(const double* qtile, int lane, const double* mu_s, const double* mf_tab, double (&p)[4]) {
    if constexpr (T1 > T0) {
        constexpr int RS = 130;
        const int j = lane & 15, g = lane >> 4;
        d4_t acc[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[b] = d4_t{0.0, 0.0, 0.0, 0.0};
        // (within a row block the steps' coordinates 16 J + 4 s run on in fours - (J, 3) -> (J + 1, 0) included.  The operands of step
        //  t + 1 are requested in front of step t's four MFMAs and nothing moves across the fence behind them: left to itself the
        //  scheduler requested all 18 matrix operands and 72 LDS values of the unrolled loop up front - 170 to 247 VGPRs, one
        //  workgroup per CU)
        constexpr int K0 = 16 * mf128_J(C0 + T0) + 4 * mf128_S(C0 + T0);
        const double* tp = mf_tab + (C0 + T0) * 64 + lane;
        const double* qk = qtile + j * RS + K0 + g;
        const double* mp = mu_s + K0 + g;
        double am = *tp, bq[4];
        {
            const double mk = *mp;
#pragma unroll
            for (int b = 0; b < 4; ++b) bq[b] = qk[16 * b * RS] - mk;
        }
#pragma unroll
        for (int t = T0; t < T1; ++t) {
            double am_n = 0.0, bq_n[4] = {0.0, 0.0, 0.0, 0.0};
            if (t + 1 < T1) {
                tp += 64; qk += 4; mp += 4;
                am_n = *tp;
                const double mk = *mp;
#pragma unroll
                for (int b = 0; b < 4; ++b) bq_n[b] = qk[16 * b * RS] - mk;
            }
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(am, bq[b], acc[b], 0, 0, 0);
            asm volatile("" ::: "memory");
            am = am_n;
#pragma unroll
            for (int b = 0; b < 4; ++b) bq[b] = bq_n[b];
        }
        double mr[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) mr[r] = mu_s[16 * I + g + 4 * r];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const double* qw = qtile + (16 * b + j) * RS + 16 * I + g;            // rows g, g + 4, g + 8, g + 12 of walker 16 b + j
#pragma unroll
            for (int r = 0; r < 4; ++r) p[b] = fma(qw[4 * r] - mr[r], acc[b][r], p[b]);
        }
    }
}
template <int WI>
__device__ __forceinline__ double mf128_wave(const double* qtile, int lane, const double* mu_s, const double* mf_tab) {
    constexpr int SPW = 18, C0 = SPW * WI;
    constexpr int IA = mf128_I(C0), IC = mf128_I(C0 + SPW - 1), IB = (IA + 1 < IC) ? IA + 1 : IC;     // this wave's row blocks, ascending
    static_assert(IC - IA <= 2, "a wave's 18 steps span at most three row blocks");
    constexpr int EA = (mf128_base(IA + 1) - C0 < SPW) ? mf128_base(IA + 1) - C0 : SPW;              // steps [0, EA) belong to IA
    constexpr int EB = (IB == IA) ? EA : ((mf128_base(IB + 1) - C0 < SPW) ? mf128_base(IB + 1) - C0 : SPW);
    double p[4] = {0.0, 0.0, 0.0, 0.0};
    mf128_segment<C0, 0, EA, IA>(qtile, lane, mu_s, mf_tab, p);
    mf128_segment<C0, EA, EB, IB>(qtile, lane, mu_s, mf_tab, p);
    mf128_segment<C0, EB, SPW, IC>(qtile, lane, mu_s, mf_tab, p);
    const int g = lane >> 4;
    double psel = 0.0;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const double v = sum_rows_f64(p[b]);
        psel = (g == b) ? v : psel;                                      // (lane group g keeps block g's walkers: lane l = walker l)
    }
    return psel;
}
__device__ __forceinline__ void like_tile_mf128(const double* qtile, double* srow, int lane, int wv, const double* mu_s, const double* prec_sym_p) {
    double part = 0.0;
    const double* tab = prec_sym_p + MF128_OFF;
    switch (wv) {
#define HENS_MF_CASE(WI) case WI: part = mf128_wave<WI>(qtile, lane, mu_s, tab); break;
        HENS_MF_CASE(0) HENS_MF_CASE(1) HENS_MF_CASE(2) HENS_MF_CASE(3)
        HENS_MF_CASE(4) HENS_MF_CASE(5) HENS_MF_CASE(6) HENS_MF_CASE(7)
#undef HENS_MF_CASE
        default: break;
    }
    srow[wv * TILE + lane] = part;
}

// Phase C of the two production kernels: every wave's partial sum into s_part[wave][walker].  Dense Gaussian at D = 32 (8 waves):
// the phase is bound by LDS bandwidth, not by FP64 issue - every wave reads its walker's whole centred row (64 lanes x 256 B per
// wave, 16 waves per CU: 2 048 clocks of the CU's 128 B / clock) to use it for 2 of the 16 row pairs.  Half the waves now read
// the row and each computes TWO of the eight partial sums, the very sums (same pairs, same order) the eight waves computed:
// same bits, half the LDS traffic.
template <int DT, int LIKE, int NW, bool CEN>
__device__ __forceinline__ void like_partials(const double* qtile, double* s_part, int lane, int wv, bool inbox, const double* mu_p,
                                              const double* prec_p, const double* prec_sym_p, double rosen_a, double rosen_b, const MfRegs& mf) {
    if constexpr (like_mf<DT, LIKE, NW>()) {          // (walkers outside the prior box ride along: a walker is a column of the product)
        if constexpr (DT == 64) like_tile_mf64<CEN>(qtile, s_part, lane, wv, mu_p, mf);
        else if constexpr (DT == 32) like_tile_mf32<CEN>(qtile, s_part, lane, wv, mu_p, mf);
        else like_tile_mf128(qtile, s_part, lane, wv, mu_p, prec_sym_p);
        return;
    }
#ifndef HENS_NO_LIKE_PAIR
    if constexpr (LIKE == LIKE_DENSE && DT == 32 && NW == 8 && CEN) {
        if (wv >= NW / 2) return;
        double pa = 0.0, pb = 0.0;
        if (inbox) {
            typedef const __attribute__((address_space(4))) double* cptr_t;
            // (opaque to the compiler: left alone it treats the pointer as free to re-read from the kernarg segment, lets the wide
            //  coefficient loads in flight overwrite its registers, and puts a dependent scalar load - kernarg -> pointer -> coefficients -
            //  in front of every batch: 16 of them in k_split1_pt<32>'s ISA, round 5)
            cptr_t psym = (cptr_t)(uintptr_t)prec_sym_p;
            asm volatile("" : "+s"(psym));
            const double* qrow = qtile + lane * (DT + 2);
            double qreg[DT];
#pragma unroll
            for (int k = 0; k < DT; k += 2) {
                const double2 v = *reinterpret_cast<const double2*>(qrow + k);
                qreg[k] = v.x; qreg[k + 1] = v.y;
            }
            switch (wv) {
#define HENS_SYM_PAIR(WI) case WI: pa = sym_quad<DT, NW, WI>(qreg, psym); pb = sym_quad<DT, NW, WI + NW / 2>(qreg, psym); break;
                HENS_SYM_PAIR(0) HENS_SYM_PAIR(1) HENS_SYM_PAIR(2) HENS_SYM_PAIR(3)
#undef HENS_SYM_PAIR
                default: break;
            }
        }
        s_part[wv * TILE + lane] = pa;
        s_part[(wv + NW / 2) * TILE + lane] = pb;
        return;
    }
#endif
    s_part[wv * TILE + lane] = like_partial<DT, LIKE, NW, CEN>(qtile, lane, wv, inbox, mu_p, prec_p, prec_sym_p, rosen_a, rosen_b);
}

// ---------------------------------------------------------------------------------------------
// Fast path for power-of-two row widths (D = 8, 16, 32, 64): same five phases, but
//   * every row chunk a thread will touch is loaded up front (NPASS x 2 x 16 B per thread in
//     flight) - the kernel is latency-bound at config-2 size, memory-level parallelism buys time;
//   * copying launches (StretchArgs::inplace == 0): the old row goes to its new home right after the gathers (most
//     proposals are rejected), the accept test only adds the accepted rows; in-place launches (hens_step on one GPU)
//     write the accepted rows only, where the walker's row already is; barriers order LDS only, so stores stay in
//     flight;
//   * scalar-load (SGPR) precision rows beat both an LDS copy and a v_readlane broadcast (measured);
//   * measured and rejected: a single fused launch for both halves (write-through rows + per-rung
//     flags: polling + sc1 traffic cost more than the boundary), and a row-resident variant that keeps
//     q in the 16 lanes of its row and rotates it with v_mov_dpp row_ror against pre-rotated precision
//     rows (no LDS tile, no barriers: correct, but 20 % slower - 16x redundant scalar loads, 1 wave/SIMD);
//   * optionally (ad_on) the ladder adaptation that follows the previous PT cascade is folded in:
//     every workgroup reduces the cascade's per-workgroup swap counts while its row gathers are
//     in flight and recomputes the ladder in one wavefront; workgroup (0,0) publishes it.  That
//     removes a dependent single-workgroup launch (~5.5 us + boundary) from every iteration.
// ---------------------------------------------------------------------------------------------
// PIPE: the context is a rank of the ladder pipeline (guest rows, flag waits, boundary-rung publishing);
// compiled out of the single-GPU instantiation, where those hooks cost ~6 % at config 2.
// PER: the context has periodic parameters (hens_set_periodic).  An instantiation of its own: as a run-time branch in the
// proposal phase it cost the D = 64 kernel 6 VGPRs and 20 % (26.8 -> 32.2 us per launch at 8 x 16384 x 64).
template <int DT, int LIKE, int MODE, int NW, bool PIPE, bool PER = false>
__global__ __launch_bounds__(NW * 64) void k_stretch_fast(const StretchArgs A) {
    constexpr bool EVAL = MODE == MODE_EVAL, MH = MODE == MODE_MH;
    constexpr bool CEN = like_centred(LIKE, DT) && !MH;   // (the MH launch draws its normals in phase B: it lost 4 us with it)
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
    double* s_beta = s_part + NW * TILE;                                 // [128] adapted ladder (ad_on)
    int32_t* s_rs = reinterpret_cast<int32_t*>(s_beta + 128);            // [TILE]
    int32_t* s_rc = s_rs + TILE;
    int32_t* s_dst = s_rc + TILE;
    int32_t* s_flag = s_dst + TILE;                                      // bit0 inbox, bit1 keep, bit2 valid
    unsigned* s_cnt = reinterpret_cast<unsigned*>(s_flag + TILE);        // [128] swap counts (ad_on)
    double* s_mu = reinterpret_cast<double*>(s_cnt + 128);               // [D] D = 64 / 128 dense: mu for the matrix-pipe phase C (mf_lds_extra)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    if constexpr (like_mf<DT, LIKE, NW>()) {
        if (tid >= NW * 64 - DT / 2) *reinterpret_cast<double2*>(s_mu + 2 * (tid - (NW * 64 - DT / 2))) = *reinterpret_cast<const double2*>(A.mu + 2 * (tid - (NW * 64 - DT / 2)));
    }
    // (D = 32: the matrix operands of phase C are requested HERE - six doubles per lane.  In front of the barrier before phase C, as at
    //  D = 64, they queue behind the workgroup's row gathers and phase C waits for them: 1 400 cycles where the MFMAs need 400)
    MfRegs mfr = like_prefetch<DT == 32 ? DT : 0, LIKE, NW>(lane, wv, A.prec_sym);
    constexpr int ADW = 1;                      // the wave that runs the early ladder adaptation (a 9th, adaptation-only
                                                // wave was measured: two 9-wave workgroups do not pack onto one CU)
    constexpr int PUSHW = 3;                    // pipeline rank, workgroup (0,0): the wave that publishes the last sweep's swap counts
    int bx = blockIdx.x, tl = blockIdx.y;
    if (A.xcd_shift > 0) {                      // dispatch order deals consecutive workgroups round-robin to the 8 XCDs
        const int sh = A.xcd_shift - 1, L = bx + (tl << sh), g = (L & 7) * ((int)(gridDim.x * gridDim.y) >> 3) + (L >> 3);
        tl = g >> sh; bx = g & ((1 << sh) - 1);
    }
    const int W = A.W;
    const int Ns = (EVAL || MH) ? W : (A.ns_x ? A.ns_x : (A.split == 0 ? A.N0 : W - A.N0));
    const int s_off = (EVAL || MH) ? 0 : (A.ns_x ? A.soff_x : (A.split == 0 ? 0 : A.N0));
    const int k0 = bx * TILE;
    const bool ad_on = !EVAL && NW >= 2 && A.ad_on;
    // mode 2: only workgroup (0,0) reduces the counts and adapts; everyone else reads its rung's beta from the ring
    const bool ad_lead = ad_on && A.ad_on == 2;
    const bool ad_here = ad_on && (!ad_lead || (blockIdx.x == 0 && blockIdx.y == 0));
#define HENS_TRACE(i) do { if (A.trace && tid == 0) A.trace[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 + (i)] = trace_stamp(); } while (0)
    HENS_TRACE(0);
#ifdef HENS_DEV_BUILD
    if (HENS_CUT_S == 9 && !EVAL && !PIPE && A.inplace) return;
    // HENS_CUT_S 10..13 (round 5, tools/cut_phase_a.sh): ONE wave's chain in front of the first barrier, the others leave at once -
    // 10: wave 0 up to its record load landed; 11: wave 0's whole phase A (record, Philox call, zz, log u); 12: wave 2 (Philox call,
    // complement's column, its row out of the compact table); 13: the two adaptation waves (count rows, ratios -> exp / reciprocals)
    if (HENS_CUT_S >= 10 && HENS_CUT_S <= 13 && !EVAL && !PIPE && A.inplace) {
        const bool mine = HENS_CUT_S <= 11 ? wv == 0 : (HENS_CUT_S == 12 ? wv == 2 : (wv == 1 || wv == 3));
        if (!mine) return;
    }
#endif
#ifdef HENS_DEV_BUILD
    if (HENS_CUT_S == 8 && !EVAL && !PIPE && A.inplace) {          // a launch of known length: every workgroup spins 6 us
        const long long t0 = wall_clock64();
        while (wall_clock64() - t0 < 600) __builtin_amdgcn_s_sleep(4);
        return;
    }
#endif
    if (PIPE && !EVAL && blockIdx.x == 0 && blockIdx.y == 0) {
        if (A.rt_flag && tid == 0) pipe_raise(A.rt_flag, A.rt_value);
        // The last sweep's swap counts (reference's schedule: due in THIS launch) go to every rank's mailbox on a wave of their own -
        // PUSHW, idle until the first barrier on a pipeline rank: its store round trip is not the adapting wave's business, which
        // sums the same rows for itself; the rows are cleared behind the first barrier, when both have read them.
        if (A.cnt_push == 2 && wv == PUSHW) pipe_push_counts(A.cp_rows, A.cp_nblocks, A.cp_np, A.cp_boxes, A.cp_nranks, A.cp_rank, A.cp_T,
                                                             A.rung_begin, W, DT, A.cp_sweep, lane, false);
    }
    // Round 5: the swap counts are waited for by the ONE wave that needs them - the adapting wave of workgroup (0,0), below - and no
    // longer by the whole workgroup at the head of the launch (rounds 3-4: wave 0 spun for every rank's counts, then a
    // __syncthreads(), then the ~4 000-cycle adaptation chain in front of the first barrier; the workgroup started its own tile
    // 4-5 us late and the launch ended with it).  The rows' flag (PF_ROWS_TOP) is everybody's: nobody gathers before it.
    const bool cnt_wait = PIPE && !EVAL && ad_here && ad_lead && (A.wmask >> PF_CNT0) != 0ull;
#ifdef HENS_DEV_BUILD      // (latency injection: dev builds only - two more live scalars cost the production kernel SGPR spills)
    unsigned long long inject_until = 0;
    if (PIPE && !EVAL && A.inject_c64 > 0 && cnt_wait) inject_until = __builtin_amdgcn_s_memtime() + (unsigned long long)A.inject_c64 * 64ull;
#else
    constexpr unsigned long long inject_until = 0;
#endif
    if (PIPE && !EVAL && A.wmask) {          // rows of the previous sweep (ladder pipeline)
        if (wv == 0 && ((A.wmask >> lane) & 1ull) && (lane < PF_CNT0 || (!ad_lead && ad_here)))
            pipe_spin(A.wflags + lane, lane >= PF_CNT0 ? A.wtarget_cnt : A.wtarget, A.wbudget, A.flags,
                      A.wstats ? A.wstats + (lane >= PF_CNT0 ? 2 : 0) : nullptr);
        // Wave 0 waits; the others meet it at the first barrier: nothing in front of that barrier touches a row (records, the
        // compact row table and the guests' home rows are this rank's own previous launch's).  Rounds 3-4 had a __syncthreads()
        // here - with its wait for EVERY outstanding load, the kernel arguments' included, at the head of every workgroup of
        // every launch whose rank has a neighbour: measured with the injection hook (which sets the count bits on a lone rank) at
        // 0.9 us per round of workgroups, 1.9 us per launch at 8 x 16384 x 64.  (D = 32 keeps it: without it that instantiation
        // spilled to scratch memory.)
        if constexpr (DT == 32) __syncthreads();
    }

    // ladder adaptation in one wavefront (tempering.py:563-596), T <= 128: lane l owns rungs l and l + 64
    // Two parts, so that a wave can run the first one (ratios, dS, exp, deltaT: ~2500 cycles of dependent FP64 divides
    // that need the counts but not the cumulative sum) while it would otherwise idle before the first barrier, and only
    // the second one (cumsum, reciprocals, update) in the shadow of the row gathers.
    const bool ad_early = ad_here && A.ad.nblocks <= 8 * A.ad.row_groups;
    // (a pipeline rank's adapting workgroup runs the whole chain in one piece - in front of the first barrier if every rank's counts
    //  are there by then, else behind its row gathers, cnt_late below: the other workgroups wait for the ring it publishes, and split
    //  around the barrier the second part's state stayed live across the gathers - measured, round 5: SGPR spills 4 -> 21 at D = 64,
    //  every workgroup's phase A 0.4 us longer)
    const bool ad_defer = ad_early && !PIPE && !ad_lead;
    const bool ad_x = ad_defer && NW >= 4 && A.ad.T <= 64 && A.ad.moving;    // (see ADX below)
    double ad_c0 = 0.0, ad_c1 = 0.0, ad_dT0 = 0.0, ad_dT1 = 0.0, ad_b0n = 1.0, ad_b1n = 1.0, ad_bb0 = 1.0, ad_bb1 = 1.0, ad_inv0 = 1.0;
    auto adapt_part1 = [&](const double cnt0, const double cnt1, const double ad_b, const double ad_b1, const bool exp_elsewhere = false) {
        const int T = A.ad.T;
        const int e0 = lane, e1 = lane + 64;
        ad_c0 = cnt0; ad_c1 = cnt1; ad_bb0 = ad_b; ad_bb1 = ad_b1;
        if (!A.ad.moving) return;
        if (exp_elsewhere) {                 // (T <= 64) the ratio chain - cnt / W, dS, exp - runs on wave ADX; here: 1 / beta differences
            const double inv0v = 1.0 / ad_b;
            ad_inv0 = inv0v;
            ad_b0n = __shfl_down(ad_b, 1);
            const double inv0n = __shfl_down(inv0v, 1);
            ad_dT0 = (e0 + 2 < T) ? inv0n - inv0v : 0.0;                           // :578; times exp(dS) after the barrier
            ad_dT1 = 0.0;
            return;
        }
        const bool two = T > 64;                                                   // wave-uniform: rungs 64.. exist
        const double r0 = cnt0 / (double)A.ad.W, r1 = two ? cnt1 / (double)A.ad.W : 0.0;   // :587
        const double kappa = A.ad.kappa;                                           // :571-572 (host)
        // ONE reciprocal per rung: 1 / beta of the next rung is the next lane's (round 3: the chain had 1 / b twice per lane, and
        // 1 / b[0] once more in the second part - on the path of every launch since the first barrier comes earlier)
        const double inv0v = 1.0 / ad_b, inv1v = two ? 1.0 / ad_b1 : 1.0;
        ad_inv0 = inv0v;
        // the value of the NEXT rung (e + 1): lane 63's successor is rung 64 = lane 0's second element
        // (lane exchanges are LDS crossbar trips: the second rung set's only for ladders above 64 rungs - wave-uniform)
        const double r0d = __shfl_down(r0, 1), b0d = __shfl_down(ad_b, 1), i0d = __shfl_down(inv0v, 1);
        double r0n = r0d, b0n = b0d, inv0n = i0d, r1n = 0.0, b1n = 1.0, inv1n = 1.0;
        if (two) {
            const double r1first = __shfl(r1, 0), b1first = __shfl(ad_b1, 0), i1first = __shfl(inv1v, 0);
            if (lane == 63) { r0n = r1first; b0n = b1first; inv0n = i1first; }
            r1n = __shfl_down(r1, 1); b1n = __shfl_down(ad_b1, 1); inv1n = __shfl_down(inv1v, 1);
        }
        ad_b0n = b0n; ad_b1n = b1n;
        double dT0 = 0.0, dT1 = 0.0;
        if (e0 + 2 < T) {
            const double dS = kappa * (r0 - r0n);                                  // :575
            dT0 = inv0n - inv0v;                                                   // :578  1 / b[e+1] - 1 / b[e]
            dT0 *= exp(dS);
        }
        if (two && e1 + 2 < T) {
            const double dS = kappa * (r1 - r1n);
            dT1 = inv1n - inv1v;
            dT1 *= exp(dS);
        }
        ad_dT0 = dT0; ad_dT1 = dT1;
    };
    auto adapt_part2 = [&]() {
        const int T = A.ad.T;
        const int e0 = lane, e1 = lane + 64;
        const double cnt0 = ad_c0, cnt1 = ad_c1, ad_b = ad_bb0, ad_b1 = ad_bb1;
        double bnew0 = ad_b, bnew1 = ad_b1;
        if (A.ad.moving) {
            const double dT0 = ad_dT0, dT1 = ad_dT1, b0n = ad_b0n, b1n = ad_b1n;
            double cs0 = 0.0, cs1 = 0.0;                                           // np.cumsum: left-to-right
            for (int i = 0; i + 2 < T; ++i) {
                const double v = i < 64 ? readlane_f64(dT0, i) : readlane_f64(dT1, i - 64);
                if (i == 0) { cs0 = v; cs1 = v; }
                else {
                    if (i <= e0) cs0 = cs0 + v;
                    if (i <= e1) cs1 = cs1 + v;
                }
            }
            const double inv0 = readlane_f64(ad_inv0, 0);                          // 1 / b[0]
            const double bn0 = 1.0 / (cs0 + inv0), bn1 = T > 64 ? 1.0 / (cs1 + inv0) : 0.0;   // :580, belong to rungs e + 1
            const double upd0 = b0n + (bn0 - b0n), upd1 = b1n + (bn1 - b1n);      // :583,:593
            const double up0 = __shfl_up(upd0, 1);
            if (e0 >= 1 && e0 + 1 < T) bnew0 = up0;
            if (T > 64) {                                                          // (wave-uniform: the second rung set)
                const double up1 = __shfl_up(upd1, 1), upd0last = readlane_f64(upd0, 63);
                if (e1 + 1 < T) bnew1 = lane >= 1 ? up1 : upd0last;
            }
        }
        if (e0 < T) s_beta[e0] = bnew0;
        if (e1 < T) s_beta[e1] = bnew1;
        if (ad_lead) {       // publish: agent-scope stores (other XCDs read them with agent-scope loads); retire the slot after next
            double* slot = A.ad_ring + (size_t)(A.ad_serial & 3u) * T;
            double* clear = A.ad_ring + (size_t)((A.ad_serial + 2u) & 3u) * T;
            if (e0 < T) {
                __hip_atomic_store(clear + e0, -1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(slot + e0, bnew0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (e1 < T) {
                __hip_atomic_store(clear + e1, -1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(slot + e1, bnew1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        if (blockIdx.x == 0 && blockIdx.y == 0) {
            // (workgroup (0,0)'s books: the three pointers are read from the kernarg segment HERE on one GPU - as by-value arguments
            //  they sit in SGPRs from the entry of every workgroup's every wave, late_kernarg)
            constexpr size_t AD = offsetof(StretchArgs, ad);
            double* const betas_out_l = PIPE ? A.ad.betas_out : late_kernarg<double*>(AD + offsetof(AdaptArgs, betas_out));
            double* const swaps_last_l = PIPE ? A.ad.swaps_last : late_kernarg<double*>(AD + offsetof(AdaptArgs, swaps_last));
            double* const swaps_total_l = PIPE ? A.ad.swaps_total : late_kernarg<double*>(AD + offsetof(AdaptArgs, swaps_total));
            if (e0 < T) wt_store(&betas_out_l[e0], bnew0);
            if (e1 < T) wt_store(&betas_out_l[e1], bnew1);
            if (e0 < T - 1 && !ad_x) {             // (ad_x: wave ADX, the only reader of the counts then, keeps these books)
                wt_store(&swaps_last_l[e0], cnt0);
                // (a pipeline rank: an atomic without a return value - `+=` is a load this wave, in front of workgroup (0,0)'s first
                //  barrier there, waits a memory round trip for: the rank's first launch 9.4 -> 8.8 us at 16 x 4096 x 32; the counts are
                //  integers, the sum is the same double.  One GPU keeps `+=`: in the gathers' shadow it costs nothing, and the atomic made
                //  config 2 0.1 us SLOWER - 17.40 -> 17.50, four alternations)
                if constexpr (PIPE) atomicAdd(&swaps_total_l[e0], cnt0);
                else wt_store(&swaps_total_l[e0], swaps_total_l[e0] + cnt0);
            }
            if (e1 < T - 1) {
                wt_store(&swaps_last_l[e1], cnt1);
                if constexpr (PIPE) atomicAdd(&swaps_total_l[e1], cnt1);
                else wt_store(&swaps_total_l[e1], swaps_total_l[e1] + cnt1);
            }
        }
    };
    auto adapt_publish = [&](const double cnt0, const double cnt1, const double ad_b, const double ad_b1) {
        adapt_part1(cnt0, cnt1, ad_b, ad_b1);
        adapt_part2();
    };

    // The counts are already reduced (one row: a pipeline rank's mailbox): wave 1 of the adapting workgroup
    // adapts right away, while wave 0 fetches the draws, so the new ladder is in the ring long before anyone asks.
    // the same workgroup pushes the last sweep's swap counts to every rank (uses the count-reduction machinery below,
    // which a pipeline rank's adaptation - counts already reduced - leaves idle)
    const bool cnt_push = PIPE && !EVAL && NW >= 2 && A.cnt_push == 1 && blockIdx.x == 0 && blockIdx.y == 0;
    const bool red_on = (ad_here && !ad_early) || cnt_push;
    // a handful of rows (one: a pipeline rank's mailbox; SWAP_ACC_ROWS: the fused half-step + cascade launch, which
    // accumulates them with atomics): all loads in flight at once, no LDS, no barrier.  The adaptation itself is a
    // ~4000-cycle dependent chain of FP64 divides and exps on one wave: a rank of the pipeline runs it right away (its
    // ring feeds the other workgroups), otherwise (every workgroup adapts for itself) the wave issues its row gathers
    // first and adapts while they are in flight.
    unsigned ad_u0[8], ad_u1[8];
    double ad_bi0 = 1.0, ad_bi1 = 1.0;
    // (column-ordered records, measured at config 2 in workgroup cycles up to the second barrier: both parts in front of the
    //  first barrier 13 070, both in the gathers' shadow 11 520, the split as it is 9 900)
    constexpr bool ad_defer_all = false;
    // Round 3: the first part was the last to reach the first barrier (4 460 cycles after the workgroup's start; the complement
    // rows' wave 2 830, the rest < 2 000): its two independent chains run on two waves - ratios -> dS -> exp on wave ADX,
    // reciprocals of the ladder on wave ADW - and meet through LDS after the barrier (ladders of up to 64 rungs).
    constexpr int ADX = 3;
    double* s_exp = s_part;                  // [64] exp(dS) per rung (phase C overwrites it after the second barrier)
    if (ad_x && wv == ADX) {
        const int T = A.ad.T, NR = A.ad.nblocks;
        const int G = A.ad.row_groups, P2 = 64 / G, p = lane & (P2 - 1), g = lane / P2;
        unsigned u[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) u[r] = (r * G + g < NR && p < T - 1) ? A.ad.swap_part[(size_t)(r * G + g) * (T - 1) + p] : 0u;
        __builtin_amdgcn_s_waitcnt(0x0F70);
        unsigned s0 = 0;
#pragma unroll
        for (int r = 0; r < 8; ++r) s0 += u[r];
        for (int m = P2; m < 64; m <<= 1) s0 += __shfl_xor(s0, m);
        const double r0 = (double)s0 / (double)A.ad.W;                             // :587
        const double r0n = __shfl_down(r0, 1);
        s_exp[lane] = (lane + 2 < T) ? exp(A.ad.kappa * (r0 - r0n)) : 1.0;        // :575
        // This wave is the ONLY reader of the count rows (the reciprocal chain on wave ADW needs none of them: every workgroup
        // reading the same 15 cache lines twice made those loads ~3 000 cycles long and wave ADW the last at the first
        // barrier, 4 150 cycles after the start against the complement wave's 3 000), so workgroup (0,0)'s books are kept here.
        if (blockIdx.x == 0 && blockIdx.y == 0) {
            if (g == 0 && p < T - 1) {
                constexpr size_t AD = offsetof(StretchArgs, ad);
                wt_store(&late_kernarg<double*>(AD + offsetof(AdaptArgs, swaps_last))[p], (double)s0);
                double* const tot = late_kernarg<double*>(AD + offsetof(AdaptArgs, swaps_total));
                wt_store(&tot[p], tot[p] + (double)s0);
            }
            if (A.ad.zero_after) {
#pragma unroll
                for (int r = 0; r < 8; ++r)
                    if (r * G + g < NR && p < T - 1 && u[r]) wt_store(&A.ad.swap_part[(size_t)(r * G + g) * (T - 1) + p], 0u);
            }
            if (A.ad.zero_rows)
                for (int e = lane; e < NR * (T - 1); e += 64) wt_store(&A.ad.zero_rows[e], 0u);
        }
    }
    auto adapt_early = [&]() {
        const int T = A.ad.T, NR = A.ad.nblocks;
        unsigned s0 = 0, s1 = 0;
#pragma unroll
        for (int r = 0; r < 8; ++r) { s0 += ad_u0[r]; s1 += ad_u1[r]; }
        const int G = A.ad.row_groups, P2 = 64 / G;
        for (int m = P2; m < 64; m <<= 1) s0 += __shfl_xor(s0, m);       // (the lane groups' partial sums)
        if (blockIdx.x == 0 && blockIdx.y == 0 && !ad_x) {
            if (A.ad.zero_after) {                   // sole reader (mode 2 / a pipeline rank): clear what was read
                const int p = lane & (P2 - 1), g = lane / P2;
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    if (r * G + g < NR && p < T - 1 && ad_u0[r]) wt_store(&A.ad.swap_part[(size_t)(r * G + g) * (T - 1) + p], 0u);
                    if (r < NR && lane + 64 < T - 1 && ad_u1[r]) wt_store(&A.ad.swap_part[(size_t)r * (T - 1) + lane + 64], 0u);
                }
            }
            if (A.ad.zero_rows)                      // every workgroup reads the rows: clear the buffer of the NEXT sweep
                for (int e = lane; e < NR * (T - 1); e += 64) wt_store(&A.ad.zero_rows[e], 0u);
        }
        adapt_part1((double)s0, (double)s1, ad_bi0, ad_bi1, ad_x);
    };
    unsigned* s_late = reinterpret_cast<unsigned*>(s_part);      // [l] late, [64 + l] own sums, [128 + l] / [192 + l] counts of rungs l / l + 64
    double* s_lateb = s_part + 128;                              // [l] / [64 + l] the ladder
    // the chain of a pipeline rank's adapting wave (below): ONE instance per kernel - behind the first barrier, in the shadow of the
    // wave's row gathers, at D = 128 (one workgroup per CU there anyway: registers to spare; eight passes of gathers hide the chain;
    // and config 5's shard is ONE round of workgroups - the launch ends with its slowest); right behind the loads, in front of the
    // first barrier, otherwise.  Measured (round 5, LABNOTES 10.1): at D = 64 the kernel sits at 128 VGPRs and behind the barrier
    // wave 0's phase-A values are live across the chain's temporaries - 129, one workgroup per CU; at D = 32 the gathers (2 us) are
    // shorter than loads + chain (3 us) and the ring came later for everybody's accept phase: 22.4 against 21.4 us per iteration
    // at 16 x 4096 x 32.  In front of the barrier the workgroup's tile starts ~2 us late - or when the counts arrive.
    constexpr bool PIPE_CHAIN_IN_SHADOW = DT == 128;
    auto pipe_chain = [&]() {
        const int T = A.ad.T, rb = A.rung_begin, np = A.cp_np;
        const bool own_acc = A.cnt_push == 2;
        unsigned m0 = s_late[128 + lane], m1 = s_late[192 + lane];
        const unsigned own = s_late[64 + lane];
        const double b0 = s_lateb[lane], b1 = s_lateb[64 + lane];
        if (s_late[lane] != 0u) {
            // a rank's counts were not there at the first look: wait for them now (the polls queue up behind this wave's own
            // gathers - loads return in order - so the chain overlaps the OTHER waves' gathers only)
            if (lane >= PF_CNT0 && lane != PF_CNT0 + A.cp_rank && ((A.wmask >> lane) & 1ull))
                pipe_spin(A.wflags + lane, A.wtarget_cnt, A.wbudget, A.flags, A.wstats ? A.wstats + 2 : nullptr, inject_until);
#ifdef HENS_DEV_BUILD
            if (inject_until)                    // (a lone rank has no flag to wait for: the injected delay alone)
                while ((long long)(__builtin_amdgcn_s_memtime() - inject_until) < 0) __builtin_amdgcn_s_sleep(2);
#endif
            if (!(own_acc && A.cp_nranks == 1)) {
                if (lane < T - 1) m0 = A.ad.swap_part[lane];
                if (lane + 64 < T - 1) m1 = A.ad.swap_part[lane + 64];
            }
        }
        if (own_acc) {                           // my pairs out of my own sums: global pair e = rung_begin + local pair
            const int j0 = lane - rb, j1 = lane + 64 - rb;
            const unsigned o0 = (unsigned)__shfl((int)own, (j0 >= 0 && j0 < np) ? j0 : 0);
            const unsigned o1 = (unsigned)__shfl((int)own, (j1 >= 0 && j1 < np) ? j1 : 0);
            if (j0 >= 0 && j0 < np) m0 = o0;
            if (j1 >= 0 && j1 < np) m1 = o1;
        }
        adapt_publish((double)m0, (double)m1, b0, b1);
    };
    // A pipeline rank on the reference's schedule: every rank's swap counts of the sweep that just ended are due in THIS launch
    // (tempering.py:563-649 adapts after the sweep), but beta is first consumed in the accept phase (red_blue.py:285-308).  The
    // adapting wave (workgroup (0,0)) only LOADS in front of the first barrier - one round trip in the shadow of wave 0's phase A:
    // its OWN rank's counts summed straight out of the accumulation rows (the same loads as the publishing wave's: it does not wait
    // for its own push to come back through the mailbox), the ladder, ONE look at the other ranks' count flags and, if they are up,
    // their counts - and runs the ~4 000-cycle chain behind the barrier, in the shadow of its row gathers: the workgroup's tile is
    // not held up by it (rounds 3-4: everything in front of the barrier; the launch ended with this workgroup, 2.3 us after the
    // others at 16 x 4096 x 32).  If a rank's counts are still on their way it comes back for them there (late).  What crosses the
    // barrier waits in LDS (s_part: nobody touches it before the second barrier) - as live registers across the gathers it cost
    // every workgroup of the launch spilled SGPRs.
    if constexpr (PIPE) {
        if (ad_early && ad_lead && wv == ADW) {
            const int T = A.ad.T;
            // (one round trip: ladder and flags are requested first, the accumulation rows last - loads return in order, the wait for
            //  the rows' sum covers all of them)
            const double b0 = (lane < T) ? A.ad.betas_in[lane] : 1.0, b1 = (lane + 64 < T) ? A.ad.betas_in[lane + 64] : 1.0;
            uint32_t fl = 0xFFFFFFFFu;
            if (cnt_wait && lane >= PF_CNT0 && lane != PF_CNT0 + A.cp_rank && ((A.wmask >> lane) & 1ull))
                fl = __hip_atomic_load(A.wflags + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            unsigned own = 0;                        // lane p: this rank's count of local pair p (cnt_push == 2)
            if (A.cnt_push == 2) own = acc_rows_sum(A.cp_rows, A.cp_nblocks, A.cp_np, lane);
            bool late = false;
            if (cnt_wait) {
                bool here = fl >= A.wtarget_cnt;
                if (inject_until && (long long)(__builtin_amdgcn_s_memtime() - inject_until) < 0) here = false;
                late = __ballot(!here) != 0ull;
                if (A.inject_c64 < 0) late = true;       // (HENS_PIPE_FORCE_LATE=1, tests: every adaptation takes the late path)
                __atomic_signal_fence(__ATOMIC_SEQ_CST);          // (acquire side: see pipe_spin)
            }
            unsigned m0 = 0, m1 = 0;                 // (row_groups = 1: the mailbox's reduced counts, one row)
            if (!late && !(A.cnt_push == 2 && A.cp_nranks == 1)) {
                if (lane < T - 1) m0 = A.ad.swap_part[lane];
                if (lane + 64 < T - 1) m1 = A.ad.swap_part[lane + 64];
            }
            s_late[lane] = late ? 1u : 0u;           // (every lane its own word: a value one lane stores for the others needs a barrier -
                                                     //  without one the compiler may read before the store, and did)
            s_late[64 + lane] = own; s_late[128 + lane] = m0; s_late[192 + lane] = m1;
            s_lateb[lane] = b0; s_lateb[64 + lane] = b1;
            if constexpr (!PIPE_CHAIN_IN_SHADOW) pipe_chain();
        }
    }
    if (ad_early && wv == ADW && !(PIPE && ad_lead)) {
        const int T = A.ad.T, NR = A.ad.nblocks;
        // (row_groups G > 1 - ladders of at most 64 / G pairs: lane = (group g, pair p), group g sums rows g, g + G, ...)
        const int G = A.ad.row_groups, P2 = 64 / G, p = lane & (P2 - 1), g = lane / P2;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            ad_u0[r] = (!ad_x && r * G + g < NR && p < T - 1) ? A.ad.swap_part[(size_t)(r * G + g) * (T - 1) + p] : 0u;
            ad_u1[r] = (!ad_x && r < NR && lane + 64 < T - 1) ? A.ad.swap_part[(size_t)r * (T - 1) + lane + 64] : 0u;
        }
        if (lane < T) ad_bi0 = A.ad.betas_in[lane];
        if (lane + 64 < T) ad_bi1 = A.ad.betas_in[lane + 64];
        // Wait for these loads HERE (the wave has nothing else to do before the first barrier), with the builtin the
        // compiler's wait-count pass understands: otherwise it guards the deferred computation with a wait that also
        // covers the row gathers issued in between, and the adaptation no longer overlaps them.
        __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0), expcnt / lgkmcnt untouched
        // first part now: this wave has nothing else to do before the barrier - unless the barrier comes early (column-ordered
        // records: phase A is one coalesced load), then everything waits for the shadow of the row gathers
        if (!ad_defer_all) adapt_early();
        if (!ad_defer) adapt_part2();
    }

    // ---- phase A (wave 0): indices and draws -------------------------------------------------------
    double factors = 0.0, lu = 0.0, Lold = 0.0, Pold = 0.0, beta_pre = 1.0;
    int own = 0;
    int32_t own_row = 0, ghome_row = 0;      // fused pipeline iteration: the walker's row (< 0: guest) and that guest's home row
    double late_ua = -1.0;                   // column-ordered records: >= 0 - (D - 1) log zz is still to be taken (phase B's shadow)
    uint32_t acc_old = 0;                    // record mode, stretch move: the slot's accept counter rides in its record
    bool valid = false;
    if (wv == 0) {
        const int k = k0 + lane;
        valid = k < Ns;
        double zz = 1.0;
        int rs = 0, rc = 0;
        if (!EVAL && A.tempered && !ad_on) beta_pre = A.betas[A.rung_begin + tl];   // off the phase-D critical path
        if (valid) {
            if (EVAL) {
                own = k;
                rs = A.loc[tl * W + own];
                rc = rs;
            } else if (MH) {                     // every walker proposes; no partner, no Hastings factor
                own = k;
                rs = A.loc[tl * W + own];
                rc = rs;
                lu = A.mh_step ? A.dr.lu[(size_t)tl * W + own]
                               : mh_log_uniform(A.mh_seed, A.mh_iter, (uint32_t)(A.rung_begin + tl) * (uint32_t)W + (uint32_t)own);
                if (PIPE && A.ghome) ghome_row = A.ghome[rs < 0 ? ~rs : 0];   // (fused pipeline iteration: every guest goes home here)
                if (A.wrec) {
                    const double2 lp = *reinterpret_cast<const double2*>(&A.wrec[tl * W + own].L);
                    Lold = lp.x; Pold = lp.y;
                } else {
                    Lold = A.L[tl * W + own];
                    Pold = A.P[tl * W + own];
                }
            } else if (A.ikeys && A.col) {
                // column-ordered records: the walker at this place is record place_column(half, place) of its rung - no round
                // key, no permutation; its row comes out of the rung's table (LDS, phase B) or with the record
                own = place_column(A.split, k, A.hb_shift);
                const WalkerRec* o = A.wrec + (tl * W + own);
                const double2 lp = *reinterpret_cast<const double2*>(&o->L);
                const int2 la = *reinterpret_cast<const int2*>(&o->loc);
#ifdef HENS_DEV_BUILD
                if (HENS_CUT_S == 10 && !PIPE && A.inplace) { s_rs[lane] = la.x; s_zz[lane] = lp.x + lp.y; return; }
#endif
                const StretchDraw sd = stretch_draw(A.iseed, A.iiter, (uint32_t)(A.rung_begin + tl) * (uint32_t)W + (uint32_t)(s_off + k));
                zz = draw_zz(sd.uz, A.ia);
                lu = log(sd.ua);                                 // red_blue.py:294 (this wave is not the last at the barrier)
                late_ua = 1.0;                                   // ((D - 1) log zz: after the row gathers have been issued)
                rs = la.x;
                acc_old = (uint32_t)la.y;                        // (consumed in phase D: the record load gates no barrier then)
                if (PIPE && A.ghome) ghome_row = A.ghome[la.x < 0 ? ~la.x : 0];
                Lold = lp.x; Pold = lp.y;
            } else if (A.ikeys) {
                // in registers: the Philox call first (it needs no key: the scalar load of the rung's round keys is in
                // flight), then the walker at this place -> its record, then the logarithms while that load is in flight
                const uint32_t* kp = A.ikeys + (size_t)(A.rung_begin + tl) * 8;
                const uint32_t key[8] = {kp[0], kp[1], kp[2], kp[3], kp[4], kp[5], kp[6], kp[7]};
                const int q = s_off + k;
                const StretchDraw sd = stretch_draw(A.iseed, A.iiter, (uint32_t)(A.rung_begin + tl) * (uint32_t)W + (uint32_t)q);
                own = (int)prp((uint32_t)place_column(A.split, k, A.hb_shift), key, A.idx_bits, (uint32_t)W);
                const WalkerRec* o = A.wrec + (tl * W + own);
                const double2 lp = *reinterpret_cast<const double2*>(&o->L);
                // ({row, accept counter} as ONE 8-byte load: a third scattered load instruction on this wave cost the
                //  iteration 0.7 us - the phase is bound by the number of memory instructions, not by bytes)
                const int2 la = *reinterpret_cast<const int2*>(&o->loc);
                rs = la.x; acc_old = (uint32_t)la.y;
                if (PIPE && A.ghome) ghome_row = A.ghome[rs < 0 ? ~rs : 0];   // (consumed in phase D: no wait in front of the barrier)
                const DrawRec dv = draw_values(own, 0, sd.uz, sd.ua, A.ia, A.ndim_active);
                zz = dv.zz; factors = dv.fac; lu = dv.lu;
                Lold = lp.x; Pold = lp.y;
            } else {
                const size_t di = (size_t)tl * W + s_off + k;
                own = A.dr.own[di];
                const int cw = A.dr.cw[di];
                zz = A.dr.zz[di];
                factors = A.dr.fac[di];
                lu = A.dr.lu[di];
                // (row indices from the compact by-field array also in record mode, where k_split1_pt keeps it current: a
                // rung's 4 W bytes stay in L2, its 32 W bytes of records would be fetched once per XCD for 4 of them)
                rs = A.loc[tl * W + own];
                // copying scheme: split 1's complement walkers were all rewritten by split 0, their row is their home
                rc = (A.split == 1 && !A.inplace) ? A.home_off + tl * W + cw : A.loc[tl * W + cw];
                if (A.wrec) {
                    const WalkerRec* o = A.wrec + (tl * W + own);
                    Lold = o->L; Pold = o->P; acc_old = o->acc;
                } else {
                    Lold = A.L[tl * W + own];
                    Pold = A.P[tl * W + own];
                }
            }
        }
        s_zz[lane] = zz;
        s_rs[lane] = rs;
        own_row = rs;
        if (!(MODE == MODE_STRETCH && A.ikeys)) s_rc[lane] = rc;
        s_dst[lane] = A.inplace ? rs : A.home_off + tl * W + own;
        s_flag[lane] = valid ? 4 : 0;
    } else if (MODE == MODE_STRETCH && A.ikeys && A.col && wv == 2) {
        const int k = k0 + lane;                 // column-ordered records: the complement is a COLUMN of the rung's row table
        int rc = 0;
        if (k < Ns) {
            const StretchDraw sd = stretch_draw(A.iseed, A.iiter, (uint32_t)(A.rung_begin + tl) * (uint32_t)W + (uint32_t)(s_off + k));
            const int colc = place_column(1 - A.split, stretch_index(sd.r22, W >> 1), A.hb_shift);
            rc = A.loc[tl * W + colc];
        }
        s_rc[lane] = rc;
    } else if (MODE == MODE_STRETCH && A.ikeys && wv == 2) {
        // the complement's row on a wave of its own (same Philox call, the other half of its output): two dependent chains
        // {draw -> walker -> record} side by side instead of one after the other
        const int k = k0 + lane;
        int rc = 0;
        if (k < Ns) {
            const uint32_t* kp = A.ikeys + (size_t)(A.rung_begin + tl) * 8;
            const uint32_t key[8] = {kp[0], kp[1], kp[2], kp[3], kp[4], kp[5], kp[6], kp[7]};
            const StretchDraw sd = stretch_draw(A.iseed, A.iiter, (uint32_t)(A.rung_begin + tl) * (uint32_t)W + (uint32_t)(s_off + k));
            const int cw = (int)prp((uint32_t)place_column(1 - A.split, stretch_index(sd.r22, W >> 1), A.hb_shift), key, A.idx_bits, (uint32_t)W);
            rc = A.loc[tl * W + cw];
        }
        s_rc[lane] = rc;
    } else if (red_on && wv == 1) {
        s_cnt[lane] = 0;
        s_cnt[lane + 64] = 0;
    }
#ifdef HENS_TRACE_WAVES      // dev builds: slot w of the trace = arrival of wave w at the first barrier
    if (A.trace && lane == 0 && wv >= 1 && wv < 8) A.trace[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 + wv] = trace_stamp();
    if (A.trace && tid == 0) {           // (thread 0 leaves: no later stamp overwrites these; the launch's results are garbage)
        A.trace[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 + 7] = trace_stamp();
        lds_barrier();
        return;
    }
#endif
#ifdef HENS_DEV_BUILD
    if (HENS_CUT_S >= 11 && HENS_CUT_S <= 13 && !EVAL && !PIPE && A.inplace) {     // (the selected wave's chain ends here: its values are used)
        asm volatile("" ::"v"(ad_dT0), "v"(ad_inv0), "v"(lu), "v"(Lold));
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        return;
    }
#endif
    HENS_TRACE(1);
    lds_barrier();
    HENS_TRACE(2);
#ifdef HENS_DEV_BUILD
    if (HENS_CUT_S == 1 && !EVAL && !PIPE && A.inplace) return;
#endif

    // ---- phase B: lanes over d, all loads first -------------------------------------------------
    const int jl = tid & (LPR - 1);
    const int rsub = tid / LPR;
    const double* __restrict__ pool_r = A.pool;
    double2 sreg[NPASS], creg[NPASS];
    bool rv[NPASS];
    // D = 128 (8 passes: 64 VGPRs of rows in flight per thread, 145 in all - ONE workgroup per CU): the gathers go in two halves of
    // four passes, each consumed (proposal -> LDS tile) before the next is requested, so the rows of one half reuse the registers
    // of the other: 106 VGPRs, two workgroups per CU, and a CU has as many bytes in flight as before (config 5's shard 28.4 -> 26.1
    // us per iteration).  Not on a pipeline rank (phase E stores the old rows from these registers there).  The pass bodies are
    // macros so that the other widths keep their one loop each, token for token: as lambdas the D = 32 launch was 0.2 us slower.
    // (D = 64 too - four passes, two halves.  Measured: 8 x 16384 x 64 with the DIAGONAL likelihood, first launch 16.27 -> 15.40 us
    //  (one box, one event-timed run of each build); the dense launch unchanged (21.9 / 21.75 us).  Registers of the diagonal
    //  launch 100 -> 84, compiler-reported room 4 -> 5 waves per SIMD - which is still TWO eight-wave workgroups per CU, so
    //  occupancy is not the explanation; the cause of the 0.9 us has not been isolated.  Rosenbrock was not measured with this
    //  change alone.)
    constexpr int HP = (DT >= 64 && !PIPE) ? NPASS / 2 : NPASS;
    constexpr int MHG = NPASS >= 2 ? RPP : 0;             // (MH normals: rows of the tile that share a Philox call, mh_normal_quad)
    static_assert(!MH || (MHG == mh_pair_rows(DT) && (HP % 2 == 0 || NPASS == 1)), "k_mh_draw pairs the rows this launch does");
    u4 mh_d{0u, 0u, 0u, 0u};
    // the proposal's step scale of this lane's two coordinates, requested once in front of the passes
    double mh_s0 = 0.0, mh_s1 = 0.0;
    if (MH && !A.mh_step) {
        mh_s0 = A.mh_kind == MH_ISO ? A.mh_scale[0] : A.mh_scale[jl * 2];
        mh_s1 = A.mh_kind == MH_ISO ? mh_s0 : A.mh_scale[jl * 2 + 1];
    }
    // the rows' indices of ALL passes out of LDS first (clamped row, no branch in between): read inside `if (rv[p])` every pass waited
    // for its own LDS round trip before it could request its rows
    int rs_i[NPASS], rc_i[NPASS];
#pragma unroll
    for (int p = 0; p < NPASS; ++p) {
        const int rr = p * RPP + rsub < TILE ? p * RPP + rsub : 0;
        rs_i[p] = s_rs[rr];
        rc_i[p] = (MH || EVAL) ? 0 : s_rc[rr];
    }
#define HENS_GATHER_PASS(p, DRAW) \
        const int r = p * RPP + rsub;                                                                                                                          \
        rv[p] = (r < TILE) && (k0 + r < Ns);   /* (= wave 0's `valid` of the row, s_flag bit 2, without the LDS read in front of every pass) */             \
        sreg[p] = double2{0.0, 0.0};                                                                                                                           \
        creg[p] = double2{0.0, 0.0};                                                                                                                           \
        if (MH && DRAW && (!MHG || (p & 1) == 0))         /* the call of this row and (MHG) of the lane's next pass' row, MHG rows on */                       \
            mh_d = mh_normal_quad(A.mh_seed, A.mh_iter, (uint32_t)(A.rung_begin + tl), (uint32_t)W, (uint32_t)(k0 + r), (uint32_t)jl);                         \
        if (rv[p]) {                                                                                                                                           \
            sreg[p] = *reinterpret_cast<const double2*>(pool_r + (PIPE ? row_off(rs_i[p], D, A.guest_delta) : (int64_t)rs_i[p] * D) + jl * 2);             \
            if (MH) {                                                                                                                                          \
                if (!DRAW) {                                                                                                                                   \
                    creg[p] = *reinterpret_cast<const double2*>(A.mh_step + ((size_t)tl * W + k0 + r) * D + jl * 2);                                           \
                } else { /* a Box-Muller pair per lane and row - the two coordinates it owns; one Philox call per TWO rows (mh_normal_quad) */                 \
                    const double2 z = (MHG && (p & 1)) ? mh_normal_from32(mh_d.z, mh_d.w) : mh_normal_from32(mh_d.x, mh_d.y);                                 \
                    creg[p] = double2{mh_s0 * z.x, mh_s1 * z.y};                                                                                               \
                }                                                                                                                                              \
            }                                                                                                                                                  \
            else if (!EVAL) creg[p] = *reinterpret_cast<const double2*>(pool_r + (PIPE ? row_off(rc_i[p], D, A.guest_delta) : (int64_t)rc_i[p] * D) + jl * 2); \
        }                                                                                                                                                     
    // (MH: the choice between the caller's steps and draws in place is made OUTSIDE the passes - as a branch inside a pass, both
    //  sides writing the same registers, the compiler guarded the draw's products with a vmcnt(0) for the other side's load, which also
    //  waits for the row load the pass has just issued: the passes' rows came one memory round trip after the other - phase B 19 600
    //  cycles at 32 x 8192 x 128 against the stretch half-step's 11 100 for twice the rows; seen in the ISA, round 5.  The draws stay
    //  BETWEEN the row requests: all requests first, then all draws, was slower - 124.9 against 121.2 us per iteration at config 5)
    const bool mh_draw = MH && !A.mh_step;
    if (mh_draw) {
#pragma unroll
        for (int p = 0; p < HP; ++p) {
            HENS_GATHER_PASS(p, 1)
        }
    } else {
#pragma unroll
        for (int p = 0; p < HP; ++p) {
            HENS_GATHER_PASS(p, 0)
        }
    }
    const double2 lov = *reinterpret_cast<const double2*>(A.lo + jl * 2);
    const double2 hiv = *reinterpret_cast<const double2*>(A.hi + jl * 2);
    double2 muv = double2{0.0, 0.0};
    if (CEN) muv = *reinterpret_cast<const double2*>(A.mu + jl * 2);
    double2 qkeep[NPASS];                          // the proposal itself stays here for phase E (the tile holds q - mu)
    if (MODE == MODE_STRETCH && wv == 0 && late_ua >= 0.0)         // the Hastings factor's logarithm, while the row gathers fly
        factors = ((double)A.ndim_active - 1.0) * log(s_zz[lane]);                // stretch.py:223
    if constexpr (PIPE) {
        if constexpr (PIPE_CHAIN_IN_SHADOW) {
            if (ad_early && ad_lead && wv == ADW) pipe_chain();
        }
        // (the accumulation rows the publishing and the adapting wave have read: cleared behind the first barrier)
        if (!EVAL && A.cnt_push == 2 && A.cp_zero && wv == PUSHW && blockIdx.x == 0 && blockIdx.y == 0)
            acc_rows_clear(const_cast<uint32_t*>(A.cp_rows), A.cp_nblocks, A.cp_np, lane);
    }
    if (ad_defer && wv == ADW) {                   // second part: the working waves' row gathers are in flight
        if (ad_defer_all) adapt_early();
        if (ad_x && lane + 2 < A.ad.T) ad_dT0 *= s_exp[lane];                      // :578-579 (the other wave's half of the first part)
        adapt_part2();
    }
    unsigned adv[8];
    double ad_b = 1.0, ad_b1 = 1.0;       // ladder values of rungs lane and lane + 64
    if (red_on) {                                  // the cascade's per-workgroup swap counts: <= 8 per thread
        const uint32_t* rows = cnt_push ? A.cp_rows : A.ad.swap_part;
        const int total = cnt_push ? A.cp_nblocks * A.cp_np : A.ad.nblocks * (A.ad.T - 1);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int e = tid + q * NT;
            adv[q] = (e < total) ? rows[e] : 0u;
        }
        if (!cnt_push && A.ad.zero_after) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int e = tid + q * NT;
                if (e < total && adv[q]) wt_store(&A.ad.swap_part[e], 0u);
            }
        }
        if (cnt_push && A.cp_zero) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int e = tid + q * NT;
                if (e < total && adv[q]) const_cast<uint32_t*>(A.cp_rows)[e] = 0u;
            }
        }
        if (!cnt_push && wv == 1 && lane < A.ad.T) ad_b = A.ad.betas_in[lane];
        if (!cnt_push && wv == 1 && lane + 64 < A.ad.T) ad_b1 = A.ad.betas_in[lane + 64];
    }
#define HENS_PROPOSE_PASS(p) \
        const int r = p * RPP + rsub;                                                                           \
        bool ok = true, finite = true;                                                                          \
        if (rv[p]) {                                                                                            \
            double2 qv;                                                                                         \
            if (EVAL) {                                                                                         \
                qv = sreg[p];                                                                                   \
            } else {                                                                                            \
                const double zz = s_zz[r];                                                                      \
                if (MH) {                                                                                       \
                    qv.x = sreg[p].x + creg[p].x; /* gaussian.py:166-167 */                                     \
                    qv.y = sreg[p].y + creg[p].y;                                                               \
                } else {                                                                                        \
                    qv.x = creg[p].x - (creg[p].x - sreg[p].x) * zz; /* stretch.py:143,145 */                   \
                    qv.y = creg[p].y - (creg[p].y - sreg[p].y) * zz;                                            \
                }                                                                                               \
                if (PER) { /* periodic parameters (see periodic_diff) */                                        \
                    const double2 pv = *reinterpret_cast<const double2*>(A.period + jl * 2);                    \
                    if (!MH) { /* stretch.py:136-145 */                                                         \
                        qv.x = creg[p].x - periodic_diff(sreg[p].x, creg[p].x, pv.x) * zz;                      \
                        qv.y = creg[p].y - periodic_diff(sreg[p].y, creg[p].y, pv.y) * zz;                      \
                    }                                                                                           \
                    qv.x = periodic_wrap(qv.x, pv.x); /* stretch.py:149-154, gaussian.py:110-115 */             \
                    qv.y = periodic_wrap(qv.y, pv.y);                                                           \
                }                                                                                               \
            }                                                                                                   \
            ok = (qv.x >= lov.x) && (qv.x <= hiv.x) && (qv.y >= lov.y) && (qv.y <= hiv.y);                      \
            finite = (fabs(qv.x) < INFINITY) && (fabs(qv.y) < INFINITY);                                        \
            if (CEN) qkeep[p] = qv;                                                                             \
            *reinterpret_cast<double2*>(qtile + r * RS + jl * 2) = double2{qv.x - muv.x, qv.y - muv.y};         \
        /* write the OLD row to its new home now (78 % of proposals are rejected at D = 32); phase E */         \
        /* overwrites only accepted rows, so the store tail after the accept test is short */                   \
            if (!EVAL && !A.inplace) {                                                                          \
                if (PIPE && tl == A.sys_rung) store_row16_sys(A.pool + (size_t)s_dst[r] * D + jl * 2, sreg[p]); \
                else store_row16(A.pool + (size_t)s_dst[r] * D + jl * 2, sreg[p]);                              \
            }                                                                                                   \
        }                                                                                                       \
        const unsigned long long bad = __ballot(!ok); /* prior.py:80-88, row-wide AND */                        \
        const unsigned long long nonfin = __ballot(!finite);                                                    \
        const int gshift = lane & ~(LPR - 1);                                                                   \
        const unsigned long long gmask = (LPR == 64) ? ~0ull : (((1ull << (LPR & 63)) - 1ull) << gshift);       \
        if (jl == 0 && rv[p]) {                                                                                 \
            if ((bad & gmask) == 0ull) atomicOr(&s_flag[r], 1);                                                 \
            if ((nonfin & gmask) != 0ull) atomicOr(A.flags, FLAG_NONFINITE_X);                                  \
        }                                                                                                      
#pragma unroll
    for (int p = 0; p < HP; ++p) {
        HENS_PROPOSE_PASS(p)
    }
    if constexpr (HP < NPASS) {
        if (mh_draw) {
#pragma unroll
            for (int p = HP; p < NPASS; ++p) {
                HENS_GATHER_PASS(p, 1)
            }
        } else {
#pragma unroll
            for (int p = HP; p < NPASS; ++p) {
                HENS_GATHER_PASS(p, 0)
            }
        }
#pragma unroll
        for (int p = HP; p < NPASS; ++p) {
            HENS_PROPOSE_PASS(p)
        }
    }
#undef HENS_GATHER_PASS
#undef HENS_PROPOSE_PASS
    if (red_on) {
        const int Tm1 = cnt_push ? A.cp_np : A.ad.T - 1;
#pragma unroll
        for (int q = 0; q < 8; ++q)
            if (adv[q]) atomicAdd(&s_cnt[(tid + q * NT) % Tm1], adv[q]);
    }
    if constexpr (DT != 32) mfr = like_prefetch<DT, LIKE, NW>(lane, wv, A.prec_sym);   // (the matrix operand of phase C: in flight across the barrier)
    HENS_TRACE(3);
    lds_barrier();
    HENS_TRACE(4);
#ifdef HENS_DEV_BUILD
    if (HENS_CUT_S == 2 && !EVAL && !PIPE && A.inplace) return;
#endif

    // ---- ladder adaptation (unless the adapting workgroup already did it up front) ------------------------------
    if (ad_here && !ad_early && wv == 1) {
        const int T = A.ad.T;
        adapt_publish((lane < T - 1) ? (double)s_cnt[lane] : 0.0, (lane + 64 < T - 1) ? (double)s_cnt[lane + 64] : 0.0, ad_b, ad_b1);
    }

    // ---- ladder pipeline: the sums of the last sweep's swap counts go to every rank's mailbox -----------------------
    if (cnt_push && wv == 1) {
        const int NP = A.cp_np;
        for (int e = lane; e < A.cp_nranks * NP; e += 64) {
            const int q = e / NP, j = e - q * NP;                    // local pair j+1 = global pair (rung_begin+j+1, rung_begin+j)
            const PipeBox bx = pipe_box(A.cp_boxes[q], A.cp_T, W, D);
            __hip_atomic_store(bx.counts + (size_t)(A.cp_sweep & 3u) * A.cp_T + (A.rung_begin + j), s_cnt[j], __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_SYSTEM);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane < A.cp_nranks) pipe_raise(pipe_box(A.cp_boxes[lane], A.cp_T, W, D).flags + PF_CNT0 + A.cp_rank, A.cp_sweep + 1);
    }

    // mode 2: the rung's new beta, requested now and consumed after the likelihood (phase D)
    double beta_ring = -1.0;
    const double* ring_slot = nullptr;
    // (a pipeline rank's adapting workgroup may publish late - cnt_late - and reads its own LDS copy; elsewhere it asks the ring like
    //  everybody: the condition stays the round-4 one there, token for token - one more live scalar pair cost the single-GPU
    //  instantiations SGPR spills and, at D = 128, 20 bytes of scratch)
    const bool ring_me = ad_lead && !(PIPE && ad_here);
    if (ring_me && wv == 0) {
        ring_slot = A.ad_ring + (size_t)(A.ad_serial & 3u) * A.ad.T + (A.rung_begin + tl);
        beta_ring = __hip_atomic_load(ring_slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }

    // ---- phase C: likelihood, lane per walker, precision rows split over the waves ----------------
    {
        const bool inbox = (s_flag[lane] & 1) != 0;
        like_partials<DT, LIKE, NW, CEN>(qtile, s_part, lane, wv, inbox, like_mf<DT, LIKE, NW>() ? s_mu : A.mu, A.prec, A.prec_sym, A.rosen_a, A.rosen_b, mfr);
    }
    HENS_TRACE(5);
    lds_barrier();
#ifdef HENS_DEV_BUILD
    if (HENS_CUT_S == 3 && !EVAL && !PIPE && A.inplace) return;
#endif

    // ---- phase D: accept / update (wave 0) -------------------------------------------------------
    if (wv == 0 && valid) {
        const bool inbox = (s_flag[lane] & 1) != 0;
        double acc = 0.0;
#pragma unroll
        for (int w2 = 0; w2 < like_nparts<DT, LIKE, NW>(); ++w2) acc += s_part[w2 * TILE + lane];
        double logl = inbox ? -0.5 * acc : A.fill;             // ensemble.py:1486-1513
        if (logl != logl) {                                    // red_blue.py:279-281
            logl = -1e300;
            atomicOr(A.flags, FLAG_NAN_LOGL);
        }
        const double logp = inbox ? A.logp_in : -INFINITY;     // prior.py:80-88
        const size_t gi = (size_t)tl * W + own;
        if (EVAL) {
            A.L[gi] = logl;
            A.P[gi] = logp;
        } else {
            double logP, prevP;
            if (A.tempered) {                                  // tempering.py:304-306,343-349
                double beta = beta_pre;
                if (ring_me) {                                 // workgroup (0,0) may still be adapting: wait for the value
                    if (beta_ring < 0.0) {                     // (the budget runs on the shader clock: see spin_expired)
                        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
                        while (beta_ring < 0.0) {
                            __builtin_amdgcn_s_sleep(1);
                            beta_ring = __hip_atomic_load(ring_slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            if (spin_expired(t0, 200000000LL)) { atomicOr(A.flags, FLAG_PIPE_TIMEOUT); break; }
                        }
                    }
                    beta = beta_ring;
                } else if (ad_on) {
                    beta = s_beta[A.rung_begin + tl];
                }
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
            const double newP = (fabs(logp) == INFINITY) ? 0.0 : logp;
            if (keep) {                                        // move.py:513-532
                if (A.wrec) {
                    if (PIPE || late_kernarg<int32_t>(offsetof(StretchArgs, norel))) {       // (written through: wt_store; a rank: see k_split1_pt's phase G)
                        store_row16(&A.wrec[gi].L, double2{logl, newP});
                        if (!MH) wt_store(&A.wrec[gi].acc, acc_old + 1u);
                    } else {
                        *reinterpret_cast<double2*>(&A.wrec[gi].L) = double2{logl, newP};
                        if (!MH) A.wrec[gi].acc = acc_old + 1u;
                    }
                } else {
                    A.L[gi] = logl;
                    A.P[gi] = newP;
                }
                if (MH || !A.wrec) atomicAdd(&A.accepted[gi], 1u);   // (the MH move counts in an array of its own)
                atomicOr(&s_flag[lane], 2);
            }
            if (PIPE && A.ghome && own_row < 0) {              // a guest: its row - new or old - goes to its home row (phase E)
                A.wrec[gi].loc = ghome_row;
                A.loc[gi] = ghome_row;
                s_dst[lane] = ghome_row;
                atomicOr(&s_flag[lane], 8);
            }
            if (PIPE && A.pub_lp && tl == A.Tl - 1) {                  // ladder pipeline: what the hot neighbour's bottom pair needs
                sys_store(A.pub_lp + own, keep ? logl : Lold);
                sys_store(A.pub_lp + W + own, keep ? newP : Pold);
            }
            if (!A.inplace) A.loc[gi] = s_dst[lane];
            if (A.keep_out) A.keep_out[(size_t)tl * Ns + k0 + lane] = keep ? 1 : 0;
        }
    }
    if (EVAL) return;
    HENS_TRACE(6);
    lds_barrier();
#ifdef HENS_DEV_BUILD
    if (HENS_CUT_S == 4 && !PIPE && A.inplace) return;
#endif

    // ---- phase E: accepted rows only (the old rows went out right after phase B) ------------------------
    double* __restrict__ pool_w = A.pool;
    if constexpr (PIPE) {
        // A pipeline rank: a guest goes home accepted or not, and rows a peer may pull are written at system scope (a property of
        // the launch).  Values and addresses of ALL passes first, then the stores back to back: the stores are inline asm, and the
        // compiler drains the memory pipeline (s_waitcnt vmcnt(0)) before it redefines a register an asm statement has read - with
        // one pass after the other every pass waited for the store of the pass before (1 200 cycles per workgroup at D = 64).
        // The proposal is read from the tile unconditionally and selected against sreg[p] as a VALUE (a conditional read merged
        // with sreg[p] becomes a select of addresses and puts the row registers into scratch memory).
        const bool sysw = tl == A.sys_rung || A.sys_all;
        double2 val[NPASS];
        double* dstp[NPASS];
        bool on[NPASS];
#pragma unroll
        for (int p = 0; p < NPASS; ++p) {
            const int r = p * RPP + rsub;
            on[p] = false;
            val[p] = double2{0.0, 0.0};
            dstp[p] = pool_w;
            if (!rv[p]) continue;
            const int fl = s_flag[r];
            const double2 ql = CEN ? qkeep[p] : *reinterpret_cast<const double2*>(qtile + r * RS + jl * 2);
            const double2 o = sreg[p];
            const bool acc = (fl & 2) != 0;
            val[p].x = acc ? ql.x : o.x;
            val[p].y = acc ? ql.y : o.y;
            on[p] = (fl & (2 | 8)) != 0;                       // accepted, or a guest's old row going home
            dstp[p] = pool_w + (size_t)s_dst[r] * D + jl * 2;
        }
        if (sysw) {
#pragma unroll
            for (int p = 0; p < NPASS; ++p) if (on[p]) store_row16_sys(dstp[p], val[p]);
        } else {
#pragma unroll
            for (int p = 0; p < NPASS; ++p) if (on[p]) store_row16(dstp[p], val[p]);
        }
    } else {
#pragma unroll
        for (int p = 0; p < NPASS; ++p) {
            const int r = p * RPP + rsub;
            if (!rv[p]) continue;
            if ((s_flag[r] & 2) == 0) continue;               // rejected: the old row is already in place
            const double2 qv = CEN ? qkeep[p] : *reinterpret_cast<const double2*>(qtile + r * RS + jl * 2);
            store_row16(pool_w + (size_t)s_dst[r] * D + jl * 2, qv);
        }
    }
    if (PIPE && A.pub_lp && A.pub_final && tl == A.Tl - 1)             // every walker of the rung has published: tell the neighbour
        if (pipe_last_ticket(A.pub_ticket, A.pub_target) && tid == 0) {
            __hip_atomic_store(A.pub_meta, (long long)A.home_off + (long long)tl * W, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            pipe_raise(A.pub_flag, A.pub_value);
        }
    if constexpr (!PIPE) if (late_kernarg<int32_t>(offsetof(StretchArgs, norel))) launch_end_wait();        // (no release fence on this launch's packet - see wt_store)
    HENS_TRACE(7);
#undef HENS_TRACE
}

// ---------------------------------------------------------------------------------------------
// Small utilities
// ---------------------------------------------------------------------------------------------
inline __global__ void k_gather_rows(const double* __restrict__ pool, const int32_t* __restrict__ loc,
                              double* __restrict__ dst, int64_t nrows, int D, int64_t guest_delta) {
    const int64_t total = nrows * D;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / D;
        const int d = (int)(i - r * D);
        dst[i] = pool[row_off(loc[r], D, guest_delta) + d];
    }
}

inline __global__ void k_pack_state(const double* __restrict__ L, const double* __restrict__ P, const int32_t* __restrict__ loc,
                             const uint32_t* __restrict__ accepted, WalkerRec* __restrict__ w, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        w[i] = make_wrec(L[i], P[i], loc[i], accepted[i]);
}
inline __global__ void k_unpack_state(const WalkerRec* __restrict__ w, double* __restrict__ L, double* __restrict__ P,
                               int32_t* __restrict__ loc, uint32_t* __restrict__ accepted, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const WalkerRec r = w[i];
        L[i] = r.L; P[i] = r.P; loc[i] = r.loc; accepted[i] = r.acc;
    }
}

// column-ordered records <-> by-field arrays (slot order): record c of rung t is the slot prp_t(c), keys = the round keys of
// the iteration the order belongs to ([T][8], k_plan_keys).  Pack also writes the compact row table in column order.
inline __global__ void k_pack_cols(const double* __restrict__ L, const double* __restrict__ P, const int32_t* __restrict__ loc,
                            const uint32_t* __restrict__ accepted, const uint32_t* __restrict__ keys, WalkerRec* __restrict__ w,
                            int32_t* __restrict__ loc_cols, int T, int W, int idx_bits) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < (int64_t)T * W; i += (int64_t)gridDim.x * blockDim.x) {
        const int t = (int)(i / W), c = (int)(i - (int64_t)t * W);
        uint32_t key[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) key[r] = keys[(size_t)t * 8 + r];
        const uint32_t sl = prp((uint32_t)c, key, idx_bits, (uint32_t)W);
        const size_t s = (size_t)t * W + sl;
        w[i] = make_wrec(L[s], P[s], loc[s], accepted[s], (int32_t)sl);
        loc_cols[i] = loc[s];
    }
}
inline __global__ void k_unpack_cols(const WalkerRec* __restrict__ w, const uint32_t* __restrict__ keys, double* __restrict__ L,
                              double* __restrict__ P, int32_t* __restrict__ loc, uint32_t* __restrict__ accepted, int T, int W,
                              int idx_bits) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < (int64_t)T * W; i += (int64_t)gridDim.x * blockDim.x) {
        const int t = (int)(i / W), c = (int)(i - (int64_t)t * W);
        uint32_t key[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) key[r] = keys[(size_t)t * 8 + r];
        const size_t s = (size_t)t * W + prp((uint32_t)c, key, idx_bits, (uint32_t)W);
        const WalkerRec r = w[i];
        L[s] = r.L; P[s] = r.P; loc[s] = r.loc; accepted[s] = r.acc;
    }
}

// hens_step_report: the accept mask of the call's last iteration(s) = what the counters gained since the mark (the stretch move's
// and the Gaussian move's counters: one of them moved)
inline __global__ void k_accept_mask(const uint32_t* __restrict__ acc, const uint32_t* __restrict__ mark, const uint32_t* __restrict__ acc_mh,
                                     const uint32_t* __restrict__ mark_mh, uint8_t* __restrict__ out, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        uint32_t d = acc[i] - mark[i];
        if (acc_mh) d += acc_mh[i] - mark_mh[i];
        out[i] = d > 255u ? 255u : (uint8_t)d;
    }
}

// ... the same without leaving record mode (hens_step_report's fast form): a slot's accept counter rides in its walker record
// (column order: record.slot names the slot; slot order: the record's index does), the Gaussian move's counters are an array by
// slot.  prev / prev_mh: the counters as the last report left them - updated here; out (may be null: snapshot only) by slot.
inline __global__ void k_report_mask(const WalkerRec* __restrict__ w, int colmode, const uint32_t* __restrict__ acc_fields,
                                     const uint32_t* __restrict__ acc_mh, uint32_t* __restrict__ prev, uint32_t* __restrict__ prev_mh,
                                     uint8_t* __restrict__ out, int T, int W) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < (int64_t)T * W; i += (int64_t)gridDim.x * blockDim.x) {
        uint32_t a;
        int64_t s = i;
        if (w) {
            const int4 la = *reinterpret_cast<const int4*>(&w[i].loc);      // {loc, acc, slot, -}: the record's second 16 bytes
            a = (uint32_t)la.y;
            if (colmode) s = (i / W) * W + la.z;
        } else a = acc_fields[i];
        uint32_t d = a - prev[s];
        prev[s] = a;
        if (acc_mh) {
            const uint32_t m = acc_mh[s];
            d += m - prev_mh[s];
            prev_mh[s] = m;
        }
        if (out) out[s] = d > 255u ? 255u : (uint8_t)d;
    }
}

inline __global__ void k_iota(int32_t* p, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        p[i] = (int32_t)i;
}

// parity mode: the caller's NumPy draws (stretch.py:93-99,129-132; red_blue.py:294) -> Draws
// order = [set 0 ascending | set 1 ascending | ...]: the moving set is positions [s_off, s_off + Ns), the complement list the
// reference indexes with rint is the other sets concatenated in set order (stretch.py:199: c = concatenate(c, axis=1)) = `order`
// with that range cut out
inline __global__ void k_prep_draws(const int32_t* __restrict__ order, const int64_t* __restrict__ rint,
                             const double* __restrict__ u_zz, const double* __restrict__ u_acc, Draws d,
                             int Tl, int W, int s_off, int Ns, double a, int D) {
    const int64_t n = (int64_t)Tl * Ns;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int tl = (int)(i / Ns), k = (int)(i - (int64_t)tl * Ns);
        const int own = order[(size_t)tl * W + s_off + k];
        const int r = (int)rint[i];
        const int cw = order[(size_t)tl * W + (r < s_off ? r : r + Ns)];
        make_draw(d, (size_t)tl * W + s_off + k, own, cw, u_zz[i], u_acc[i], a, D);
    }
}

// ---------------------------------------------------------------------------------------------
// Metropolis-Hastings proposals: the draws of a GaussianMove (gaussian.py:68-270).
// ---------------------------------------------------------------------------------------------
// parity mode: the caller's accept uniforms (mh.py:157) -> log
inline __global__ void k_mh_prep(const double* __restrict__ u_acc, double* __restrict__ lu, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        lu[i] = log(u_acc[i]);
}

struct MhDrawArgs {
    double* step;            // [Tl][W][D]
    double* lu;              // [Tl][W]
    const double* scale;     // MH_ISO: [1] std dev; MH_DIAG: [D] std devs; MH_FULL: [D][D] lower Cholesky factor, row-major
    uint64_t iter, seed;
    int32_t Tl, W, D, rung_begin, kind, chol_lds;
    double* dbg_u;           // debug (hens_debug_draws): the raw accept uniforms [Tl][W], or nullptr
};
// Philox mode: step = scale * z (isotropic / diagonal) or chol * z (full covariance), z ~ N(0, 1) by
// Box-Muller from Philox counters keyed (iteration, global rung, walker, coordinate pair); the accept
// uniform likewise.  One workgroup = 64 walkers of one rung; z goes through LDS for the triangular product.
inline __global__ __launch_bounds__(256) void k_mh_draw(const MhDrawArgs A) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    double* z = reinterpret_cast<double*>(smem_raw);                  // [64][D + 1]
    const int D = A.D, W = A.W, ZS = D + 1;
    double* chol = z + 64 * ZS;                                       // [D][D + 1] when A.chol_lds (odd stride: no bank conflicts)
    const int tl = blockIdx.y, w0 = blockIdx.x * 64;
    const uint32_t rung = (uint32_t)(A.rung_begin + tl);
    const int npair = (D + 1) / 2;
    if (A.kind == MH_FULL && A.chol_lds)
        for (int i = threadIdx.x; i < D * D; i += blockDim.x) chol[(i / D) * ZS + (i % D)] = A.scale[i];
    for (int i = threadIdx.x; i < 64 * npair; i += blockDim.x) {
        const int wl = i / npair, pr = i - wl * npair, w = w0 + wl;
        if (w >= W) continue;
        const double2 n2 = mh_normal_pair(A.seed, A.iter, rung, (uint32_t)W, (uint32_t)w, (uint32_t)pr, (uint32_t)mh_pair_rows(D));
        z[wl * ZS + 2 * pr] = n2.x;
        if (2 * pr + 1 < D) z[wl * ZS + 2 * pr + 1] = n2.y;
    }
    if (threadIdx.x < 64 && w0 + (int)threadIdx.x < W) {
        const int w = w0 + threadIdx.x;
        A.lu[(size_t)tl * W + w] = mh_log_uniform(A.seed, A.iter, rung * (uint32_t)W + (uint32_t)w);
        if (A.dbg_u) A.dbg_u[(size_t)tl * W + w] = mh_uniform(A.seed, A.iter, rung * (uint32_t)W + (uint32_t)w);
    }
    __syncthreads();
    const bool in_lds = A.kind == MH_FULL && A.chol_lds;
    const double* cf = in_lds ? chol : A.scale;
    const int CS = in_lds ? ZS : D;
    for (int i = threadIdx.x; i < 64 * D; i += blockDim.x) {
        const int wl = i / D, d = i - wl * D, w = w0 + wl;
        if (w >= W) continue;
        double v;
        if (A.kind == MH_FULL) {
            v = 0.0;
            for (int k = 0; k <= d; ++k) v = fma(cf[(size_t)d * CS + k], z[wl * ZS + k], v);
        } else {
            v = (A.kind == MH_ISO ? A.scale[0] : A.scale[d]) * z[wl * ZS + d];
        }
        A.step[((size_t)tl * W + w) * D + d] = v;
    }
}

// ---------------------------------------------------------------------------------------------
// Philox plan: the state-independent part of the stretch proposals, ahead of time.
// One workgroup per (iteration, resident rung): a pseudo-random BALANCED red/blue labelling
// (label(w) = prp(w) >= ceil(W/2): exactly ceil(W/2) walkers in split 0, like the reference's shuffle
// of arange(W) % 2, red_blue.py:119-124), each half listed in ASCENDING walker order like the
// reference's boolean masks - a tile of 64 moving walkers then touches ~128 consecutive ids, so the
// per-walker scalars (loc, L, P, accept counts) move as whole cache lines - and for every position
// the complement index and the stretch / accept terms (stretch.py:93-99,129-132,223; red_blue.py:294).
// One launch plans a batch of iterations; it depends on nothing but (seed, iteration), so it runs
// ahead of the stepping kernels on its own low-priority stream.
// ---------------------------------------------------------------------------------------------
struct PlanArgs {
    Draws dr;             // [NB][Tl][W] each
    uint64_t iter0;       // iteration index of the first planned iteration
    uint64_t seed;
    double a;
    int32_t Tl, W, D, rung_begin, idx_bits;
    int32_t T, cb;        // cb > 0: block-balanced labelling with cb columns per block (see block_rank); 0: label = prp >= N0
    DrawRec* rec;         // [NB][W / cb][64] the second half-step's draws in k_split1_pt's order (see DrawRec), or nullptr
    int32_t rec_only;     // with rec: the by-position arrays of the second half-step are not needed (nobody reads them)
    DrawRec* rec1;        // one-launch iteration (k_iter; with rec, W <= 65536): [NB][W / cb][64] the FIRST half-step's draws in
    DrawRec* rec3;        //   block order, and for every second-half walker the first-half draws of ITS complement (k_iter
                          //   replays that walker's first half-step itself); no by-position arrays at all
    uint32_t* keys;       // [NB][T][8] round keys of every rung's cascade column map (cb > 0), or nullptr
    double* dbg_uzz;      // debug (hens_debug_draws): the raw uniforms behind zz / lu, [NB][Tl][W], or nullptr
    double* dbg_uacc;
    int32_t nsets, nsets_pad_;   // > 2: RedBlueMove(nsplits = nsets) with device draws (k_plan_sets, one iteration at a time)
    int32_t* order;       // [Tl][W] k_plan_sets: the sets one after the other, each in ascending walker order (hens_ctx_impl::order)
};

// exclusive scan of this thread's value across the workgroup (wave shuffles + one LDS hop)
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t x, uint32_t* wtot, int tid, int nt) {
    const int lane = tid & 63, wave = tid >> 6, nw = (nt + 63) >> 6;
    uint32_t inc = x;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t y = __shfl_up(inc, off);
        if (lane >= off) inc += y;
    }
    if (lane == 63) wtot[wave] = inc;
    __syncthreads();
    if (wave == 0) {
        uint32_t t = (lane < nw) ? wtot[lane] : 0u;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t y = __shfl_up(t, off);
            if (lane >= off) t += y;
        }
        if (lane < nw) wtot[lane] = t;                     // inclusive wave totals
    }
    __syncthreads();
    return inc - x + (wave > 0 ? wtot[wave - 1] : 0u);
}

inline __global__ __launch_bounds__(1024) void k_plan(const PlanArgs A) {
    // shapes WITHOUT block-balanced labels (untempered ensembles, ladders above 64 rungs, walker counts that are not a
    // multiple of the block): label = prp(w) >= ceil(W/2), both halves listed in ascending walker order through a scan
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    __shared__ uint32_t wtot[16];
    __shared__ uint32_t skey[8];
    const int tid = threadIdx.x;
    const int nt = blockDim.x;
    const int ib = blockIdx.x / A.Tl;                      // iteration within the batch
    const int job = blockIdx.x - ib * A.Tl;                // resident rung
    const uint64_t it = A.iter0 + (uint64_t)ib;
    const int W = A.W;
    const uint32_t rung = (uint32_t)(A.rung_begin + job);
    const int N0 = (W + 1) / 2;
    int32_t* ord = reinterpret_cast<int32_t*>(smem_raw);             // [W] ordered walker ids
    uint16_t* lab = reinterpret_cast<uint16_t*>(ord + W);            // [W] 1: first half-step
    if (tid < 64) {                                                  // one wave draws the rung's round keys
        const PrpKey K = prp_key(A.seed, it, PURPOSE_SPLIT, rung);
        if (tid < 8) skey[tid] = K.k[tid];
    }
    __syncthreads();
    uint32_t key[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) key[r] = skey[r];
    for (int i = tid; i < W; i += nt) lab[i] = prp((uint32_t)i, key, A.idx_bits, (uint32_t)W) >= (uint32_t)N0 ? 0 : 1;
    __syncthreads();
    const int chunk = (W + nt - 1) / nt;                   // consecutive ids per thread
    const int lo = min(W, tid * chunk), hi = min(W, lo + chunk);
    uint32_t z = 0;
    for (int i = lo; i < hi; ++i) z += lab[i];
    uint32_t z0 = block_excl_scan(z, wtot, tid, nt);                 // first-half walkers before this thread's chunk
    uint32_t o0 = (uint32_t)lo - z0;                                 // second-half walkers before it
    for (int i = lo; i < hi; ++i) {
        if (lab[i]) ord[z0++] = i;
        else ord[N0 + o0++] = i;
    }
    __syncthreads();
    const size_t base = ((size_t)ib * A.Tl + job) * W;
    for (int p = tid; p < W; p += nt) {
        const int own = ord[p];
        const StretchDraw sd = stretch_draw(A.seed, it, rung * (uint32_t)W + (uint32_t)p);
        const bool s0 = p < N0;
        const int cw = ord[(s0 ? N0 : 0) + stretch_index(sd.r22, s0 ? W - N0 : N0)];
        store_draw(A.dr, base + p, draw_values(own, cw, sd.uz, sd.ua, A.a, A.D));
        if (A.dbg_uzz) {
            A.dbg_uzz[base + p] = sd.uz;
            A.dbg_uacc[base + p] = sd.ua;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// RedBlueMove(nsplits = n > 2) with device draws (round 4; red_blue.py:41-47,119-124,148-197): walker w of a rung carries label
// prp(w) mod n under the rung's keyed permutation - arange(W) % n shuffled: set k holds ceil((W - k) / n) walkers - every set
// is listed in ascending walker order like the reference's boolean masks, the sets one after the other (`order`), and position q
// of set k draws its complement uniformly from the OTHER sets concatenated in set order (stretch.py:93-99 on red_blue.py:
// 183-197's `sets`): order with the moving range cut out.  One workgroup per rung, one iteration per launch; the copying launches
// then run set after set exactly as the parity API's (StretchArgs::ns_x / soff_x).
// ---------------------------------------------------------------------------------------------
inline __global__ __launch_bounds__(256) void k_plan_sets(const PlanArgs A) {
    __shared__ uint32_t cnt[8][256];
    __shared__ uint32_t skey[8];
    __shared__ int s_off[9];
    const int tid = threadIdx.x, job = blockIdx.x, W = A.W, n = A.nsets;
    const uint32_t rung = (uint32_t)(A.rung_begin + job);
    if (tid < 64) {
        const PrpKey K = prp_key(A.seed, A.iter0, PURPOSE_SPLIT, rung);
        if (tid < 8) skey[tid] = K.k[tid];
    }
    if (tid == 0) {
        s_off[0] = 0;
        for (int k = 0; k < n; ++k) s_off[k + 1] = s_off[k] + (W - k + n - 1) / n;      // red_blue.py:120-124
    }
    __syncthreads();
    uint32_t key[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) key[r] = skey[r];
    const int chunk = (W + 255) / 256, lo = min(W, tid * chunk), hi = min(W, lo + chunk);
    uint32_t mine[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int w = lo; w < hi; ++w) {
        const uint32_t l = prp((uint32_t)w, key, A.idx_bits, (uint32_t)W) % (uint32_t)n;
#pragma unroll
        for (int k = 0; k < 8; ++k) mine[k] += (l == (uint32_t)k) ? 1u : 0u;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) cnt[k][tid] = mine[k];
    __syncthreads();
    if (tid < n) {                                       // set tid: where every thread's chunk starts inside the set's range
        uint32_t run = (uint32_t)s_off[tid];
        for (int t = 0; t < 256; ++t) { const uint32_t c = cnt[tid][t]; cnt[tid][t] = run; run += c; }
    }
    __syncthreads();
    int32_t* ord = A.order + (size_t)job * W;
    uint32_t at[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) at[k] = cnt[k][tid];
    for (int w = lo; w < hi; ++w) {
        const uint32_t l = prp((uint32_t)w, key, A.idx_bits, (uint32_t)W) % (uint32_t)n;
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (l == (uint32_t)k) ord[at[k]++] = w;
    }
    __threadfence_block();
    __syncthreads();
    const size_t base = (size_t)job * W;
    for (int q = tid; q < W; q += 256) {
        int k = 0;
        while (q >= s_off[k + 1]) ++k;
        const int so = s_off[k], Ns = s_off[k + 1] - so;
        const StretchDraw sd = stretch_draw(A.seed, A.iter0, rung * (uint32_t)W + (uint32_t)q);
        const int r = stretch_index(sd.r22, W - Ns);                     // stretch.py:93-99
        const int own = __builtin_nontemporal_load(ord + q), cw = __builtin_nontemporal_load(ord + (r < so ? r : r + Ns));
        store_draw(A.dr, base + q, draw_values(own, cw, sd.uz, sd.ua, A.a, A.D));
        if (A.dbg_uzz) {
            A.dbg_uzz[base + q] = sd.uz;
            A.dbg_uacc[base + q] = sd.ua;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// The plan of shapes WITH block-balanced labels (k_plan_keys -> k_plan_draws), round 3: for the launches that read their
// draws from memory (copying launches, pipeline ranks, k_iter) and for hens_debug_draws; the two-launch iteration of
// hens_step computes the same values in registers (stretch_draws_at) and needs the round keys only.
// Round 2's plan was one 1024-thread workgroup per (iteration, rung) - inverse Feistel network per walker, block rank from
// eight hashes, a scan for ascending lists, two Philox calls, ~40 us of life on a CU: run beside the stepping kernels it
// cost them 2.8 us per iteration at config 2 (a stepping workgroup that cannot start until a plan workgroup leaves its CU
// doubles its launch), run alone 4 us.  Now one thread per (iteration, rung, place), nothing ordered, nothing shared.
// ---------------------------------------------------------------------------------------------
inline __global__ void k_plan_keys(const PlanArgs A, int nb) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= nb * A.Tl) return;
    const int ib = g / A.Tl, job = g - ib * A.Tl;
    const uint32_t rung = (uint32_t)(A.rung_begin + job);
    const PrpKey K = prp_key(A.seed, A.iter0 + (uint64_t)ib, PURPOSE_PTPERM, rung);
    uint32_t* dst = A.keys + ((size_t)ib * A.T + rung) * 8;
#pragma unroll
    for (int r = 0; r < 8; ++r) dst[r] = K.k[r];
}

inline __global__ __launch_bounds__(256) void k_plan_draws(const PlanArgs A) {
    const int ib = blockIdx.y / A.Tl, job = blockIdx.y - ib * A.Tl;
    const uint32_t rung = (uint32_t)(A.rung_begin + job);
    const uint64_t it = A.iter0 + (uint64_t)ib;
    const int W = A.W, hb = A.cb >> 1, hb_shift = __ffs(hb) - 1, N0 = W >> 1;
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= W) return;
    const uint4* kp = reinterpret_cast<const uint4*>(A.keys + ((size_t)ib * A.T + rung) * 8);
    const uint4 ka = kp[0], kb = kp[1];
    const uint32_t key[8] = {ka.x, ka.y, ka.z, ka.w, kb.x, kb.y, kb.z, kb.w};
    const size_t base = ((size_t)ib * A.Tl + job) * W;
    const bool s0 = q < N0;
    const int p = s0 ? q : q - N0;
    const PlaceDraw pd = stretch_draws_at(A.seed, it, rung, q, key, W, A.idx_bits, hb_shift);
    const DrawRec rc = draw_values(pd.own, pd.cw, pd.uz, pd.ua, A.a, A.D);
    const int blk = p >> hb_shift, ml = p & (hb - 1);
    const size_t ri = ((size_t)ib * (W / A.cb) + blk) * TILE + (size_t)rung * hb + ml;
    if (A.rec1) {                                            // k_iter: everything in block order
        if (s0) {
            A.rec1[ri] = rc;
        } else {
            A.rec[ri] = rc;
            // the complement sits at place pd.r of the first half: its own draws once more (same counter -> same values)
            const PlaceDraw p1 = stretch_draws_at(A.seed, it, rung, pd.r, key, W, A.idx_bits, hb_shift);
            A.rec3[ri] = draw_values(p1.own, p1.cw, p1.uz, p1.ua, A.a, A.D);
        }
        return;
    }
    A.dr.own[base + q] = rc.own;
    if (s0 || !(A.rec && A.rec_only)) {
        A.dr.cw[base + q] = rc.cw;
        A.dr.zz[base + q] = rc.zz;
        A.dr.fac[base + q] = rc.fac;
        A.dr.lu[base + q] = rc.lu;
    }
    if (A.rec && !s0) A.rec[ri] = rc;
    if (A.dbg_uzz) {
        A.dbg_uzz[base + q] = pd.uz;
        A.dbg_uacc[base + q] = pd.ua;
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
// k_pt_chain builds the column form from the reference's draws (parity mode); in Philox mode column c
// meets slot prp_t(c) of rung t (a keyed pseudo-random matching per pair, computed inline - one
// permutation per pair has the same distribution as the reference's two).
// ---------------------------------------------------------------------------------------------
inline __global__ void k_pt_invert(const int64_t* __restrict__ iperm, int32_t* __restrict__ inv, int npairs, int W) {
    const int64_t n = (int64_t)npairs * W;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int j = (int)(i / W);
        inv[(size_t)j * W + iperm[i]] = (int32_t)(i - (int64_t)j * W);
    }
}

// thread c follows column c: rows j = 0..T-2 of iperm/i1perm are the pairs i = T-1-j.
inline __global__ void k_pt_chain(const int64_t* __restrict__ iperm, const int64_t* __restrict__ i1perm,
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
    const double* P;            // local current buffers [Tl][W]
    const int32_t* loc;
    double* Lnew;               // local next buffers
    double* Pnew;
    int32_t* locnew;
    const WalkerRec* wrec;      // whole ladder resident, inside hens_step's record mode: {L, P, loc} come from / go to the
    WalkerRec* wrecnew;         // walker records (and locnew, the compact row table), L / P / loc above are not touched
    const double* betas;        // [T]
    const int32_t* colslot;     // [T][W]
    const double* colu;         // [T-1][W] uniforms in column order (parity) or nullptr (Philox)
    uint8_t* selcol;            // [T-1][W] swap decisions in column order, row j <-> pair T-1-j (or nullptr)
    int32_t* srcfull;           // [T][W] global source slot id of the walker arriving at every slot (sharded) or nullptr
    uint32_t* swap_part;        // [nblocks][T-1] per-workgroup swap counts (reduced by the adaptation)
    uint64_t iter;
    uint64_t seed;
    unsigned long long* trace;  // debug: per-workgroup phase timestamps, or nullptr
    int32_t T, W, Tl, rung_begin, idx_bits;
    int32_t acc_rows;           // 0: swap_part is [nblocks][T-1], a row per workgroup; else a power of two: [acc_rows][T-1], accumulated (clean on entry)
};

constexpr int PT_COLS = 16;      // columns per workgroup: W/16 workgroups keep every CU busy at W = 4096
constexpr int PT_THREADS = 256;
__host__ __device__ inline int pt_threads(int T) { const int ne = (T * 16 + 63) & ~63; return ne < 256 ? 256 : (ne > 1024 ? 1024 : ne); }   // k_pt_cascade: a thread per element

// LDS per (rung, column) element: L f64, log-uniform f64, P f64, loc i32, slot i32; + betas[T] + swap bitmasks
__host__ __device__ inline size_t pt_lds_layout(int T) { return (size_t)T * PT_COLS * (8 + 8 + 8 + 4 + 4) + (size_t)T * 8 + (size_t)PT_COLS * ((T + 31) / 32) * 4; }

// One launch = the whole hot->cold cascade.  Two dependent global-load levels only
// (colslot -> {L, P, loc}); everything after that runs out of LDS.  No inter-workgroup
// communication: the swap totals needed for the ladder adaptation leave as per-workgroup rows
// and are reduced after the kernel boundary (a last-block ticket + release fence inside this
// kernel cost more than the cascade itself).
template <bool PHILOX>
__global__ __launch_bounds__(1024) void k_pt_cascade(const PtArgs A) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int T = A.T, W = A.W;
    const size_t NE = (size_t)T * PT_COLS;
    double* Lc = reinterpret_cast<double*>(smem_raw);            // [T][PT_COLS]
    double* lu = Lc + NE;                                        // [T][PT_COLS] (row j = pair T-1-j)
    double* Pc = lu + NE;                                        // [T][PT_COLS]
    double* sbeta = Pc + NE;                                     // [T]
    int32_t* locc = reinterpret_cast<int32_t*>(sbeta + T);       // [T][PT_COLS]
    int32_t* scol = locc + NE;                                   // [T][PT_COLS]
    uint32_t* smask = reinterpret_cast<uint32_t*>(scol + NE);    // [PT_COLS][MW] swap bitmask per column (bit i = pair (i, i-1))
    const int MW = (T + 31) / 32;
    const int tid = threadIdx.x;
    const int NTH = blockDim.x;                                  // pt_threads(T): a thread per element up to 1024 (round 5; 256 before)
    const int c0 = blockIdx.x * PT_COLS;
    const uint64_t it = A.iter;
#define PT_TRACE(i) do { if (A.trace && tid == 0) A.trace[(size_t)blockIdx.x * 8 + (i)] = trace_stamp(); } while (0)
    PT_TRACE(0);

    // phase 1: column slots (Philox: computed in place from the rung's keys), then everything the
    // column needs - independent gathers, the log-uniform overlaps their latency
    for (int t = tid; t < T; t += NTH) sbeta[t] = A.betas[t];
    // (record mode: the slot this thread reads in phase 1 is the slot it writes in phase 3 - same element e, destination = the
    //  element's own slot - so the slot's accept counter, which stays with the slot, is kept in a register from the record load
    //  instead of being read again in front of the store: one dependent memory round trip less at the launch's tail, round 5)
    constexpr int PT_KEEP = 2;                                   // elements per thread covered (T <= 128: NE <= 2 x 1024)
    uint32_t acc_keep[PT_KEEP] = {0u, 0u};
    for (int e = tid, k = 0; e < (int)NE; e += NTH, ++k) {
        const int t = e / PT_COLS, cc = e - t * PT_COLS, c = c0 + cc;
        if (c < W) {
            const int slot = PHILOX ? pt_slot(A.seed, it, t, T, c, A.idx_bits, W) : A.colslot[(size_t)t * W + c];
            scol[e] = slot;
            const int tl = t - A.rung_begin;
            if (A.wrec) {
                const WalkerRec wr = A.wrec[(size_t)t * W + slot];
                Lc[e] = wr.L; Pc[e] = wr.P; locc[e] = wr.loc;
                if (k == 0) acc_keep[0] = wr.acc;
                if (k == 1) acc_keep[1] = wr.acc;
            } else {
                Lc[e] = A.Lfull[(size_t)t * W + slot];
                if (tl >= 0 && tl < A.Tl) {
                    Pc[e] = A.P[(size_t)tl * W + slot];
                    locc[e] = A.loc[(size_t)tl * W + slot];
                } else {
                    locc[e] = -1;                                // walker owned by another rank
                }
            }
            if (t < T - 1) {
                const double u = PHILOX ? pt_uniform(A.seed, it, t, W, c) : A.colu[(size_t)t * W + c];
                lu[e] = log(u);                                  // tempering.py:535
            }
        }
    }
    PT_TRACE(1);
    PT_TRACE(2);
    __syncthreads();
    PT_TRACE(3);

    // phase 2: one lane per column walks hot -> cold.  Nothing is stored inside the loop (the swap
    // decisions go into a register bitmask), so the LDS reads of the next steps are issued ahead and
    // only the compare/select chain is serial.
    // (straight-line walks for the ladders of the BASELINE configs, as in k_split1_pt: round 5 - the run-time loop took 223 cycles
    //  per pair at 32 rungs against ~100 straight-line)
    auto walk = [&](auto tt) {
        constexpr int TT = decltype(tt)::value;                      // 0: run-time ladder length
        const int Tn = TT ? TT : T;
        const int cc = tid;
        double cL = Lc[(size_t)(Tn - 1) * PT_COLS + cc];
        uint32_t m = 0;
#pragma unroll
        for (int i0 = Tn - 1; i0 >= 1; i0 -= 8) {                   // 8 steps per LDS round trip
            double Lb[8], lv[8], db[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int i = (i0 - q >= 1) ? i0 - q : 1;
                Lb[q] = Lc[(size_t)(i - 1) * PT_COLS + cc];
                lv[q] = lu[(size_t)(Tn - 1 - i) * PT_COLS + cc];
                db[q] = sbeta[i - 1] - sbeta[i];                                 // tempering.py:518-522
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int i = i0 - q;
                if (i >= 1) {
                    const double pacc = db[q] * (cL - Lb[q]);                    // tempering.py:538
                    const bool sw = pacc > lv[q];                                // tempering.py:541
                    m |= sw ? (1u << (i & 31)) : 0u;
                    cL = sw ? cL : Lb[q];                        // no swap: the resident becomes the carried walker
                    if ((i & 31) == 0 || i == 1) {
                        smask[cc * MW + (i >> 5)] = m;
                        m = 0;
                    }
                }
            }
        }
    };
    if (tid < PT_COLS && c0 + tid < W) {
        if (T == 8) walk(std::integral_constant<int, 8>{});
        else if (T == 16) walk(std::integral_constant<int, 16>{});
        else if (T == 32) walk(std::integral_constant<int, 32>{});
        else walk(std::integral_constant<int, 0>{});
    }
    PT_TRACE(4);
    __syncthreads();
    PT_TRACE(5);

    // phase 3: write the permuted L / P / loc of the resident rungs straight from LDS.  The walker
    // arriving on rung t of a column comes from rung t-1 if pair (t, t-1) swapped; otherwise the
    // carried walker settles there, and it started its fall at t + (number of consecutive swapped
    // pairs directly above).
    auto bit = [&](int cc, int i) -> bool { return (i >= 1 && i < T) && ((smask[cc * MW + (i >> 5)] >> (i & 31)) & 1u); };
    for (int e = tid, k = 0; e < (int)NE; e += NTH, ++k) {
        const int t = e / PT_COLS, cc = e - t * PT_COLS, c = c0 + cc;
        if (c >= W) continue;
        int st;
        bool sel_out;
        if (MW == 1) {                                            // T <= 32: the whole column mask in one register
            const uint32_t mw = smask[cc];
            sel_out = (mw >> ((T - 1 - t) & 31)) & 1u;
            if ((mw >> t) & 1u) st = t - 1;                       // bit 0 is never set
            else st = t + __builtin_ctz(~(mw >> 1 >> t));         // consecutive swapped pairs directly above
        } else {
            sel_out = bit(cc, T - 1 - t);
            if (bit(cc, t)) {
                st = t - 1;
            } else {
                st = t;
                while (bit(cc, st + 1)) ++st;
            }
        }
        if (t < T - 1 && A.selcol) A.selcol[(size_t)t * W + c] = sel_out ? 1 : 0;   // row j <-> pair T-1-j
        const int se = st * PT_COLS + cc;
        const int dslot = scol[e];
        if (A.srcfull) A.srcfull[(size_t)t * W + dslot] = st * W + scol[se];
        const int tl = t - A.rung_begin;
        if (tl < 0 || tl >= A.Tl) continue;
        const size_t di = (size_t)tl * W + dslot;
        if (A.wrecnew) {
            // (the slot keeps its accept counter; record mode holds the whole ladder: tl = t, di = this element's own record)
            const uint32_t acc_own = (k < PT_KEEP && A.rung_begin == 0) ? (k == 0 ? acc_keep[0] : acc_keep[1]) : A.wrec[di].acc;
            A.wrecnew[di] = make_wrec(Lc[se], Pc[se], locc[se], acc_own);
            A.locnew[di] = locc[se];
            continue;
        }
        A.Lnew[di] = Lc[se];
        A.locnew[di] = locc[se];      // -1: row + log-prior arrive from another rank (hens_pt_finish_sharded)
        if (locc[se] >= 0) A.Pnew[di] = Pc[se];
    }
    for (int i = 1 + tid; i < T; i += NTH) {               // pair (i, i-1) -> swap_part index i-1
        unsigned n = 0;
        for (int cc = 0; cc < PT_COLS && c0 + cc < W; ++cc) n += bit(cc, i) ? 1u : 0u;
        // (acc_rows: accumulated with atomics into a handful of rows that the next launch's folded adaptation sums - like the fused
        //  launch's counts; else one row per workgroup, for the stand-alone adaptation)
        if (A.acc_rows) { if (n) atomicAdd(&A.swap_part[(size_t)(blockIdx.x & (A.acc_rows - 1)) * (T - 1) + (i - 1)], n); }
        else A.swap_part[(size_t)blockIdx.x * (T - 1) + (i - 1)] = n;
    }
    PT_TRACE(6);
#undef PT_TRACE
}

// ---------------------------------------------------------------------------------------------
// Second half-step + PT cascade + swap counts in ONE launch (Philox mode, block-balanced labels).
//
// One workgroup owns cb consecutive cascade columns on all T rungs: NE = cb * T = 128 slots.  Exactly
// TILE = 64 of the walkers in those slots carry label 1 (block_rank), i.e. they are this workgroup's
// share of the second red/blue half-step; the other 64 were moved by the first half-step's launch.
// After the accept test the workgroup holds the post-move (L, P, loc) of all 128 slots in LDS - exactly
// what the cascade of its columns reads - so the hot -> cold walk follows immediately, without the
// launch boundary, the ramp and the gathers of a separate cascade kernel:
//   A  thread per slot : column map (Feistel), label rank, {loc, L, P} and - for the moving walkers - their
//                        draw record by walker id; the complement of a second-half walker sits in its home row
//   B  lanes over d    : row gathers, q = c - (c - s) zz, box test by ballot, old row -> new home   (as k_stretch_fast)
//   C  lane per walker : likelihood (like_partial)
//   D  lane per walker : tempered accept test; the result goes into the cascade's LDS tables
//   E  lanes over d    : accepted rows
//   F  lane per column : the walk (k_pt_cascade phase 2)
//   G  thread per slot : permuted L / P / loc into the next buffers; swap counts by atomics into
//                        SWAP_ACC_ROWS rows (the adapting workgroup of the next launch reduces and clears them)
// ---------------------------------------------------------------------------------------------
// Rows the cascade launches accumulate their swap counts into with atomics (workgroup b: row b % rows); the adapting wave
// of the next launch sums them straight out of memory, 8 loads per lane.  Round 3: 8 rows for every ladder meant 64
// workgroups x (T - 1) atomics on ONE cache line at config 2 - 0.7 us per iteration, 0.5 of it contention; a ladder of
// at most 64 / G pairs now spreads over 8 G rows, G lane groups summing 8 rows each (acc_row_groups).
constexpr int SWAP_ACC_ROWS_MAX = 64;
__host__ __device__ inline int acc_row_groups(int T) {
    int p2 = 1;
    while (p2 < T - 1) p2 <<= 1;
    const int g = p2 >= 64 ? 1 : 64 / p2;
    return g > SWAP_ACC_ROWS_MAX / 8 ? SWAP_ACC_ROWS_MAX / 8 : g;
}

struct FusedArgs {
    double* pool;
    const WalkerRec* wrec;                                    // current buffer: the first half-step is already in
    WalkerRec* wrecnew;                                       // next buffer: after the cascade
    const int32_t* loc;                                       // [T][W] the row of every walker once more, compact (a rung's
    int32_t* locnew;                                          // 4 W bytes stay in L2): where the complements are looked up
    const double* betas;                                      // [T]
    const uint32_t* keys;                                     // [T][8] round keys of the rungs' column maps (k_plan_keys)
    const uint32_t* keys_next;                                // COL: the NEXT iteration's round keys (the order the records are written in)
    uint32_t* accepted;                                       // [T][W]
    uint32_t* swap_acc;                                       // [acc_rows][T-1]
    const double* lo; const double* hi; const double* mu; const double* prec; const double* prec_sym;
    const double* period;                                     // [D] periodic parameters (see StretchArgs::period), or nullptr
    unsigned* flags;
    unsigned long long* trace;                                // debug: 8 phase timestamps per workgroup, or nullptr
    double logp_in, fill, rosen_a, rosen_b;
    double a;                                                 // stretch scale (stretch.py:129-132)
    uint64_t iter, seed;
    int32_t T, W, idx_bits, cb, cb_shift, ndim_active;
    int32_t acc_rows;                                         // rows of swap_acc (a power of two): workgroup b adds to row b % acc_rows
    int32_t norel;                                            // see StretchArgs::norel
    // ---- PIPE instantiation: a rank of the ladder pipeline (T, keys, betas stay GLOBAL; the state arrays are the rank's) ----
    int32_t Tl, rung_begin;                                   // resident rungs [rung_begin, rung_begin + Tl)
    int32_t cbl, cbl_shift;                                   // columns per workgroup: 128 / Tl (a multiple of cb)
    int64_t guest_delta;                                      // see row_off
    int32_t* ghome;                                           // [2][2][W] home row of a guest (see StretchArgs::ghome)
    char* box; char* box_hot; char* box_cold;                 // my mailbox, the hot / cold neighbour's (or nullptr)
    const double* pool_cold;                                  // cold neighbour's walker pool (rows that move up are pulled)
    unsigned long long* stats;                                // debug wait statistics or nullptr
    long long budget;                                         // wall-clock ticks a flag wait may take
    uint32_t sweep;
    int32_t par, nranks, rank;
    int32_t sys_rows;                                         // rows are stored at system scope (a peer pulls rows out of this pool)
    int32_t no_move;                                          // PIPE: the cascade alone (the iteration's move was a full-ensemble MH launch)
};

// (pipe: the cascade tables hold one more rung - what the hot neighbour's columns carry - and the bottom boundary's lists)
__host__ __device__ constexpr size_t fused_lds_base(int D, int NW, bool pipe = false) {
    return ((size_t)TILE * (D + 2) + (size_t)NW * TILE + 5 * TILE + 3 * 2 * TILE + 64 + (pipe ? 2 * TILE : 0)) * 8 +
           (2 * 2 * TILE + 5 * TILE + 64 + (pipe ? 3 * TILE : 0)) * 4;
}
__host__ __device__ inline size_t fused_lds_bytes(int D, int NW, int like, bool pipe = false) { return fused_lds_base(D, NW, pipe) + mf_lds_extra(D, like); }

// SHORT: the ladder length does not divide 128 - cb T < 128 slots and cb T / 2 < 64 moving walkers per workgroup.  An
// instantiation of its own: with run-time bounds the full-tile launch lost its compile-time-true row guards, 0.2 us at
// config 2 (tools/ab3.sh: 22.4 / 22.6 / 22.5 us per iteration before / with run-time bounds / with this parameter).
// PIPE: the context is a rank of the ladder pipeline (round 3) - the workgroup owns 128 / Tl consecutive columns on
// the rank's Tl rungs (a whole number of label blocks: again 128 slots, 64 of them moving), publishes its share of the
// hottest rung to the hot neighbour, starts its walk from what that neighbour's columns carry, hands its own columns on to
// the cold neighbour and settles the bottom boundary - all hand-offs per column block, rows updated in place.
// COL: column-ordered records (StretchArgs::col) - wrec / loc are read in THIS iteration's column order (record c of rung t is
// the walker column c meets: the slot threads' loads are coalesced and wait for no key), and the next buffers are written in
// the NEXT iteration's column order: the slot column c meets is prp_t(c), its column next time prp_inv of that under the next
// iteration's key - a permutation and its inverse in the shadow of the record load, scattered STORES at the tail.
template <int DT, int LIKE, int NW, bool PER = false, bool SHORT = false, bool PIPE = false, bool COL = false>
__global__ __launch_bounds__(NW * 64) void k_split1_pt(const FusedArgs A) {
    static_assert(DT == 8 || DT == 16 || DT == 32 || DT == 64 || DT == 128, "power-of-two row width");
    static_assert(!(PIPE && SHORT), "pipeline ranks: full tiles");
    static_assert(!(COL && (PER || SHORT)), "column-ordered records: full tiles");
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    constexpr bool CEN = like_centred(LIKE, DT);
    constexpr int D = DT, RS = DT + 2, NT = NW * 64, LPR = DT / 2, RPP = NT / LPR, NPASS = (TILE + RPP - 1) / RPP;
    constexpr int NE = 2 * TILE;
    constexpr int NEX = PIPE ? NE + TILE : NE;            // a pipeline rank's tables hold the hot neighbour's columns on top
    static_assert(NT >= 2 * NE, "one thread per slot + one per cascade uniform");
    constexpr bool WIDE = NT >= 2 * NE + 64;              // a further wave for the ladder, else the slot threads fetch it
    double* qtile = reinterpret_cast<double*>(smem_raw);                 // [TILE][RS]
    double* s_part = qtile + TILE * RS;                                  // [NW][TILE]
    double* s_zz = s_part + NW * TILE;                                   // [TILE] per moving walker
    double* s_fac = s_zz + TILE;
    double* s_lu = s_fac + TILE;
    double* s_Lold = s_lu + TILE;
    double* s_Pold = s_Lold + TILE;
    int2* s_own = reinterpret_cast<int2*>(s_Lold);                       // [NE] {accept counter, slot index} of slot e, phase D -> G (s_Lold / s_Pold: 2 x TILE doubles, otherwise unused)
    double* Lc = s_Pold + TILE;                                          // [NEX] cascade tables, element e = t * cb + cc
    double* Pc = Lc + NEX;
    double* lupt = Pc + NEX;                                             // [NE] log-uniform of pair T-1-t on column cc
    double* sbeta = lupt + NE;                                           // [64]
    int32_t* locc = reinterpret_cast<int32_t*>(sbeta + 64);              // [NEX]
    int32_t* scol = locc + NEX;                                          // [NE] slot of the element
    int32_t* s_rs = scol + NE;                                           // [TILE]
    int32_t* s_rc = s_rs + TILE;
    int32_t* s_dst = s_rc + TILE;                                        // [TILE] (PIPE: the row an accepted / guest row is written to)
    int32_t* s_flag = s_dst + TILE;                                      // bit0 inbox, bit1 keep, bit3 guest (PIPE)
    int32_t* s_el = s_flag + TILE;                                       // [TILE] element of the moving walker
    uint32_t* smask = reinterpret_cast<uint32_t*>(s_el + TILE);          // [cb][MW] swap bitmask per column
    int32_t* s_src = reinterpret_cast<int32_t*>(smask + 64);             // PIPE [TILE] bottom boundary: row that moves down
    int32_t* s_yrow = s_src + TILE;                                      // PIPE [TILE] row (cold neighbour's pool) that moves up
    double* s_mu = reinterpret_cast<double*>(smem_raw + fused_lds_base(DT, NW, PIPE));   // [D] D = 64 / 128 dense: mu for the matrix-pipe phase C

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    if constexpr (like_mf<DT, LIKE, NW>()) {
        if (tid >= NW * 64 - DT / 2) *reinterpret_cast<double2*>(s_mu + 2 * (tid - (NW * 64 - DT / 2))) = *reinterpret_cast<const double2*>(A.mu + 2 * (tid - (NW * 64 - DT / 2)));
    }
    // (D = 32: the matrix operands of phase C are requested HERE - six doubles per lane.  In front of the barrier before phase C, as at
    //  D = 64, they queue behind the workgroup's row gathers and phase C waits for them: 1 400 cycles where the MFMAs need 400)
    MfRegs mfr = like_prefetch<DT == 32 ? DT : 0, LIKE, NW>(lane, wv, A.prec_sym);
    const int TG = A.T;                                                  // the whole ladder
    const int T = PIPE ? A.Tl : A.T;                                     // the rungs this workgroup holds
    const int W = A.W, CB = PIPE ? A.cbl : A.cb, CS = PIPE ? A.cbl_shift : A.cb_shift;
    const int R0 = PIPE ? A.rung_begin : 0;
    const bool has_top = PIPE && R0 + T < TG, has_bot = PIPE && R0 > 0;
    const int TE = T + (has_top ? 1 : 0);                                // the walk's rungs: mine + the hot neighbour's columns
    const int c0 = blockIdx.x * CB;
    const int MW = (TE + 31) >> 5;
    // ladders whose length does not divide 128: cb = the largest power of two with cb T <= 128, so a workgroup holds
    // NEr = cb T <= 128 slots and NM = NEr / 2 <= 64 moving walkers; the lanes / rows beyond them idle
    const int NEr = SHORT ? (T << CS) : 2 * TILE, NM = SHORT ? (NEr >> 1) : TILE;
#define FUSED_TRACE(i) do { if (A.trace && tid == 0) A.trace[(size_t)blockIdx.x * 8 + (i)] = trace_stamp(); } while (0)
    FUSED_TRACE(0);
#ifdef HENS_DEV_BUILD
    if (HENS_CUT_F == 9) return;
#endif
#ifdef HENS_DEV_BUILD
    if (HENS_CUT_F == 8) {
        const long long t0 = wall_clock64();
        while (wall_clock64() - t0 < 600) __builtin_amdgcn_s_sleep(4);
        return;
    }
#endif

    // ---- phase A: one thread per slot, the moving walkers' draws on waves of their own -----------------------------------
    // One memory round trip of round keys (a 32-byte load per rung that hits L2, k_plan_keys) and one of walker records in
    // front of the row gathers; no planned draws (round 3).  Slot thread e = (rung t, column cc of the block): the column
    // map gives the walker in the slot, whose {L, P, row} is one 32-byte record (the phase is bound by the number of
    // cache lines it pulls in).  The block's last cb/2 columns meet the walkers that MOVE in this half-step
    // (place_column): their slot threads keep the record for phase D and are the lanes that run it; walker m = t cb/2 +
    // (cc - cb/2) of the tile.  What is random about walker m comes from one Philox call keyed by its split position
    // (stretch_draws_at), computed twice on otherwise idle waves - one turns it into the complement's row (second Feistel
    // network, a lookup in the compact row table), one into zz / (D - 1) log zz / log u - while the round keys are in flight.
    // Program order matters in the slot threads: the column map is computed BEFORE the record loads are issued (its
    // cycle-walking loop makes the compiler wait for every load in flight), and the records of the walkers that stay
    // (cascade tables) are consumed after the row gathers have been issued.
    // (PIPE: the workgroup's columns are several label blocks of cb columns: the second half of EACH block moves.)
    WalkerRec wr_n{};                                                    // slot threads: the record of the walker in the slot
    bool stays = false;
    int slot_n = 0;
    int32_t home_n = 0;                                                  // PIPE: home row of a moving walker that sits in a guest row
    const int HS = A.cb_shift - 1, HB = 1 << HS;                         // half a label block
    const int HW = CB >> 1;                                              // moving walkers per rung of this workgroup
    const bool nomove = PIPE && A.no_move != 0;                          // every slot stays: the records go straight into the tables
    constexpr int CWW = WIDE ? 5 : 2, FLW = WIDE ? 6 : 3;               // the waves of the moving walkers' draws
    // index of the moving walker met by column cc of rung t among the workgroup's 64
    auto mover_of = [&](int t, int cc) -> int {
        if (PIPE) return (t << (CS - 1)) + ((cc >> (HS + 1)) << HS) + (cc & (HB - 1));
        return (t << (CS - 1)) + cc - HB;
    };
    if (wv == 0) s_flag[lane] = 0;                                       // (also the idle lanes of a short tile: never in the box)
    if (tid < NEr) {                                                     // (short ladders: the slots beyond cb T do not exist)
        const int e = tid, t = e >> CS, cc = e & (CB - 1), c = c0 + cc;
        if (COL) {
            // the record of the walker column c meets IS record c: coalesced, waits for no key and no permutation (where the
            // slot's new record goes - scol, phase G - is worked out by the threads of the cascade uniforms, below)
            wr_n = A.wrec[(size_t)t * W + c];
        } else {
            const uint4* kp = reinterpret_cast<const uint4*>(A.keys) + (size_t)(R0 + t) * 2;
            const uint4 ka = kp[0], kb = kp[1];
            const uint32_t key[8] = {ka.x, ka.y, ka.z, ka.w, kb.x, kb.y, kb.z, kb.w};
            slot_n = (int)prp((uint32_t)c, key, A.idx_bits, (uint32_t)W);
            wr_n = A.wrec[(size_t)t * W + slot_n];
        }
        stays = PIPE ? (nomove || ((cc >> HS) & 1) == 0) : cc < HB;
        if (!stays) {
            const int m = mover_of(t, cc);
            s_rs[m] = wr_n.loc;
            if (PIPE) home_n = A.ghome[wr_n.loc < 0 ? ~wr_n.loc : 0];    // (unconditional; consumed in phase D)
            else s_dst[m] = t * W + slot_n;                              // (walker index: the accept counters)
        }
        if (!COL) scol[e] = slot_n;
        if (!WIDE && e < TE) sbeta[e] = A.betas[R0 + e];
    }
    if (tid >= NE && tid < 2 * NE) {
        // the cascade's log-uniforms (a Philox call and a log per element: ~1700 cycles of dependent ALU) on the two
        // waves that would otherwise idle until the barrier, not in the shadow of the slot chain above
        const int e = tid - NE, t = e >> CS, c = c0 + (e & (CB - 1));
        // row t of the table: pair TE-1-t of the walk = global pair R0 + TE-1-t, whose uniform is row TG-1-(R0+TE-1-t)
        if (t < TE - 1) lupt[e] = log(pt_uniform(A.seed, A.iter, PIPE ? TG - R0 - TE + t : t, W, c));   // tempering.py:535
    }
    // Two Philox calls per mover instead of three (column order, 8 waves): complement row + zz + (D - 1) log zz on one wave, the
    // accept uniform's logarithm on a second.  What stands in front of the first barrier is ALU issue - two 8-wave workgroups per CU
    // all drawing at once - as much as the length of any one chain: a timing-only build without the draws runs this launch in 7.4
    // instead of 9.0 us; with the slot index carried in the records (one Feistel walk per slot instead of two) the last wave
    // reaches the barrier at 2 900 cycles instead of 3 400 and the launch takes 8.8 us.  (Slot order keeps the three-wave split:
    // its chains wait for round keys.)
    constexpr bool ONEDRAW = COL && WIDE;
    if (COL && WIDE && !ONEDRAW && wv == 3 && lane < NM) {       // log of the movers' accept uniforms (the same Philox call as FLW's, its other half)
        const int m = lane, t = m >> (CS - 1), q = (W >> 1) + blockIdx.x * HW + (m & (HW - 1));
        s_lu[m] = log(stretch_draw(A.seed, A.iter, (uint32_t)(R0 + t) * (uint32_t)W + (uint32_t)q).ua);   // red_blue.py:294
    }
    if (COL && (NW >= 8 ? (wv == 4 || wv == 7) : wv < 2)) {
        // where slot e's record goes (phase G): the slot column c meets is prp_t(c) under this iteration's key - carried in the
        // record since k_pack_cols computed it (WalkerRec::slot) - its column in the next iteration's order the inverse of that
        // under the next key: one Feistel walk that nothing in front of the barrier needs, on two waves with nothing else to do
        // (measured, when it was two walks: on the slot threads the barrier came 1000 cycles later, on the uniform-drawing waves
        // 700)
        const int e = (NW >= 8 ? (wv == 4 ? 0 : 64) : wv * 64) + lane, t = e >> CS, c = c0 + (e & (CB - 1));
        const uint32_t slot = (uint32_t)A.wrec[(size_t)t * W + c].slot;      // (= prp_t(c) under this iteration's key: carried in the record)
        const uint4* kn = reinterpret_cast<const uint4*>(A.keys_next) + (size_t)(R0 + t) * 2;
        const uint4 na = kn[0], nb = kn[1];
        const uint32_t keyn[8] = {na.x, na.y, na.z, na.w, nb.x, nb.y, nb.z, nb.w};
        scol[e] = (int)prp_inv(slot, keyn, A.idx_bits, (uint32_t)W);
    }
    if (WIDE && wv == (ONEDRAW ? FLW : 4)) {         // (one draw wave: the wave that drew zz has nothing else to do)
        if (lane < TE) sbeta[lane] = A.betas[R0 + lane];
    }
    if (ONEDRAW && wv == CWW && lane < NM && !nomove) {
        const int m = lane, t = m >> (CS - 1), q = (W >> 1) + blockIdx.x * HW + (m & (HW - 1));
        const StretchDraw sd = stretch_draw(A.seed, A.iter, (uint32_t)(R0 + t) * (uint32_t)W + (uint32_t)q);
        const int cw = place_column(0, stretch_index(sd.r22, W >> 1), HS);
        const int32_t rcv = A.loc[t * W + cw];                             // (in flight under the logarithms)
        const double z = draw_zz(sd.uz, A.a);
        s_zz[m] = z;
        s_fac[m] = ((double)A.ndim_active - 1.0) * log(z);                // stretch.py:223
        s_rc[m] = rcv;
    }
    if (ONEDRAW && wv == FLW && lane < NM && !nomove) {   // (the accept uniform's logarithm: the same call's other half, on a wave of its own)
        const int m = lane, t = m >> (CS - 1), q = (W >> 1) + blockIdx.x * HW + (m & (HW - 1));
        s_lu[m] = log(stretch_draw(A.seed, A.iter, (uint32_t)(R0 + t) * (uint32_t)W + (uint32_t)q).ua);   // red_blue.py:294
    }
    if (!ONEDRAW && (wv == CWW || wv == FLW) && lane < NM && !nomove) {
        const int m = lane, t = m >> (CS - 1), q = (W >> 1) + blockIdx.x * HW + (m & (HW - 1));   // split position (second half)
        const StretchDraw sd = stretch_draw(A.seed, A.iter, (uint32_t)(R0 + t) * (uint32_t)W + (uint32_t)q);
        if (wv == CWW) {
            int cw;
            if (COL) {                       // the row table is in column order: the complement's column is its index
                cw = place_column(0, stretch_index(sd.r22, W >> 1), HS);
            } else {
                const uint4* kp = reinterpret_cast<const uint4*>(A.keys) + (size_t)(R0 + t) * 2;
                const uint4 ka = kp[0], kb = kp[1];
                const uint32_t key[8] = {ka.x, ka.y, ka.z, ka.w, kb.x, kb.y, kb.z, kb.w};
                cw = (int)prp((uint32_t)place_column(0, stretch_index(sd.r22, W >> 1), HS), key, A.idx_bits, (uint32_t)W);
            }
            s_rc[m] = A.loc[t * W + cw];
        }
        if (wv == FLW || CWW == FLW) {
            if (COL && WIDE) {               // the two logarithms on two waves (one wave with both was the last at the barrier by
                const double z = draw_zz(sd.uz, A.a);                             // 600 cycles; behind the barrier they stretched phase B)
                s_zz[m] = z;
                s_fac[m] = ((double)A.ndim_active - 1.0) * log(z);                // stretch.py:223
            } else {
                const DrawRec dv = draw_values(0, 0, sd.uz, sd.ua, A.a, A.ndim_active);
                s_zz[m] = dv.zz; s_fac[m] = dv.fac; s_lu[m] = dv.lu;
            }
        }
    }
#ifdef HENS_TRACE_WAVES      // dev builds: slot w of the trace = arrival of wave w at the first barrier (slot 0: the workgroup's start)
    if (A.trace && lane == 0 && wv >= 1 && wv < 8) A.trace[(size_t)blockIdx.x * 8 + wv] = trace_stamp();
    if (A.trace && tid == 0) { lds_barrier(); return; }
#endif
    FUSED_TRACE(1);
    lds_barrier();
    FUSED_TRACE(2);
#ifdef HENS_DEV_BUILD
    if (HENS_CUT_F == 1) return;
#endif

    // ---- phase B: lanes over d, all loads first ----------------------------------------------------------
    const int jl = tid & (LPR - 1);
    const int rsub = tid / LPR;
    const double* __restrict__ pool_r = A.pool;
    double2 sreg[NPASS], creg[NPASS];
    bool rv[NPASS];
    int rs_i[NPASS], rc_i[NPASS];                   // (the rows' indices of all passes out of LDS first: see k_stretch_fast)
#pragma unroll
    for (int p = 0; p < NPASS; ++p) {
        const int rr = p * RPP + rsub < TILE ? p * RPP + rsub : 0;
        rs_i[p] = s_rs[rr];
        rc_i[p] = s_rc[rr];
    }
#pragma unroll
    for (int p = 0; p < NPASS; ++p) {
        const int r = p * RPP + rsub;
        rv[p] = r < NM && !nomove;
        sreg[p] = double2{0.0, 0.0};
        creg[p] = double2{0.0, 0.0};
        if (rv[p]) {
            sreg[p] = *reinterpret_cast<const double2*>(pool_r + (PIPE ? row_off(rs_i[p], D, A.guest_delta) : (int64_t)rs_i[p] * D) + jl * 2);
            creg[p] = *reinterpret_cast<const double2*>(pool_r + (PIPE ? row_off(rc_i[p], D, A.guest_delta) : (int64_t)rc_i[p] * D) + jl * 2);
        }
    }
    const double2 lov = *reinterpret_cast<const double2*>(A.lo + jl * 2);
    const double2 hiv = *reinterpret_cast<const double2*>(A.hi + jl * 2);
    double2 muv = double2{0.0, 0.0};
    if (CEN) muv = *reinterpret_cast<const double2*>(A.mu + jl * 2);
    double2 qkeep[NPASS];                          // the proposal itself stays here for phase E (the tile holds q - mu)
    if (tid < NEr && stays) {                       // (loaded before the row gathers were issued: it arrives before them)
        Lc[tid] = wr_n.L; Pc[tid] = wr_n.P; locc[tid] = wr_n.loc;
    }
#pragma unroll
    for (int p = 0; p < NPASS; ++p) {
        const int r = p * RPP + rsub;
        bool ok = true, finite = true;
        if (rv[p]) {
            const double zz = s_zz[r];
            double2 qv;
            qv.x = creg[p].x - (creg[p].x - sreg[p].x) * zz;             // stretch.py:143,145
            qv.y = creg[p].y - (creg[p].y - sreg[p].y) * zz;
            if (PER) {                                                   // periodic parameters: stretch.py:136-154
                const double2 pv = *reinterpret_cast<const double2*>(A.period + jl * 2);
                qv.x = periodic_wrap(creg[p].x - periodic_diff(sreg[p].x, creg[p].x, pv.x) * zz, pv.x);
                qv.y = periodic_wrap(creg[p].y - periodic_diff(sreg[p].y, creg[p].y, pv.y) * zz, pv.y);
            }
            ok = (qv.x >= lov.x) && (qv.x <= hiv.x) && (qv.y >= lov.y) && (qv.y <= hiv.y);
            finite = (fabs(qv.x) < INFINITY) && (fabs(qv.y) < INFINITY);
            if (CEN) qkeep[p] = qv;
            *reinterpret_cast<double2*>(qtile + r * RS + jl * 2) = double2{qv.x - muv.x, qv.y - muv.y};
        }
        const unsigned long long bad = __ballot(!ok);                    // prior.py:80-88, row-wide AND
        const unsigned long long nonfin = __ballot(!finite);
        const int gshift = lane & ~(LPR - 1);
        const unsigned long long gmask = (LPR == 64) ? ~0ull : (((1ull << (LPR & 63)) - 1ull) << gshift);
        if (jl == 0 && rv[p]) {
            if ((bad & gmask) == 0ull) atomicOr(&s_flag[r], 1);
            if ((nonfin & gmask) != 0ull) atomicOr(A.flags, FLAG_NONFINITE_X);
        }
    }
    // (the matrix operand of phase C - D = 64 / 128 dense: five / eighteen doubles per lane out of prec_sym - requested in front of the
    //  barrier as in k_stretch_fast, in flight across it.  Rounds 4-5 requested it behind the barrier: phase C 5 600 cycles at
    //  8 x 16384 x 64 against the first launch's 1 290, tools/trace_pipe_phases.py)
    if constexpr (DT != 32) mfr = like_prefetch<DT, LIKE, NW>(lane, wv, A.prec_sym);
    FUSED_TRACE(3);
    lds_barrier();
#ifdef HENS_DEV_BUILD
    if (HENS_CUT_F == 2) return;
#endif

    // ---- phase C: likelihood ------------------------------------------------------------------------------
    if (!nomove) {
        const bool inbox = (s_flag[lane] & 1) != 0;
        like_partials<DT, LIKE, NW, CEN>(qtile, s_part, lane, wv, inbox, like_mf<DT, LIKE, NW>() ? s_mu : A.mu, A.prec, A.prec_sym, A.rosen_a, A.rosen_b, mfr);
    }
    FUSED_TRACE(4);
    lds_barrier();
#ifdef HENS_DEV_BUILD
    if (HENS_CUT_F == 3) return;
#endif

    // ---- phase D: accept / update into the cascade's tables (the moving walkers' slot threads) ---------------------
    if (tid < NEr && !stays) {
        const int e = tid, t = e >> CS, m = mover_of(t, e & (CB - 1));
        const bool inbox = (s_flag[m] & 1) != 0;
        double acc = 0.0;
#pragma unroll
        for (int w2 = 0; w2 < like_nparts<DT, LIKE, NW>(); ++w2) acc += s_part[w2 * TILE + m];
        double logl = inbox ? -0.5 * acc : A.fill;                      // ensemble.py:1486-1513
        if (logl != logl) {                                             // red_blue.py:279-281
            logl = -1e300;
            atomicOr(A.flags, FLAG_NAN_LOGL);
        }
        const double logp = inbox ? A.logp_in : -INFINITY;              // prior.py:80-88
        const double beta = sbeta[t];
        const double Lold = wr_n.L, Pold = wr_n.P;
        double lt = logl * beta;                                        // tempering.py:304-306,343-349
        if (lt != lt) lt = -INFINITY;
        const double logP = lt + logp;
        double lo_ = Lold * beta;
        if (lo_ != lo_) lo_ = -INFINITY;
        const double prevP = lo_ + Pold;
        const double lnpdiff = s_fac[m] + logP - prevP;                 // red_blue.py:292
        const bool keep = lnpdiff > s_lu[m];                            // red_blue.py:294
        const double newP = (fabs(logp) == INFINITY) ? 0.0 : logp;      // move.py:513-532
        Lc[e] = keep ? logl : Lold;
        Pc[e] = keep ? newP : Pold;
        if (PIPE) {
            // a walker that arrived through the pipeline sits in a guest row until now: this half-step writes its row - the
            // proposal or the old one - into the pool row vacated by the walker that left in exchange (phase E)
            const bool guest = wr_n.loc < 0;
            const int32_t row = guest ? home_n : wr_n.loc;
            locc[e] = row;
            s_dst[m] = row;
            if (guest) s_flag[m] |= 8;
        } else {
            locc[e] = wr_n.loc;                                         // rows are updated in place (see StretchArgs::wrec)
        }
        if (keep) {
            wr_n.acc += 1u;                                             // (phase G writes the slot's record back)
            s_flag[m] |= 2;
        }
    }
    if (tid < NEr) s_own[tid] = int2{(int)wr_n.acc, wr_n.slot};            // (phase G's paired record stores read them from here)
    FUSED_TRACE(5);
    lds_barrier();
#ifdef HENS_DEV_BUILD
    if (HENS_CUT_F == 4) return;
#endif

    // ---- phase F: one lane per column walks hot -> cold (tempering.py:515-541) ---------------------------------
    // (a serial chain of T-1 compare / select steps; with the ladder length a compile-time constant every LDS address
    // is an immediate and the loop is straight-line code)
    auto walk = [&](auto tt) {
        constexpr int TT = decltype(tt)::value;                          // 0: runtime ladder length
        const int Tn = TT ? TT : TE;
        const int cc = lane;
        double cL = Lc[((Tn - 1) << CS) + cc];
        uint32_t m = 0;
#pragma unroll
        for (int i0 = Tn - 1; i0 >= 1; i0 -= 8) {
            double Lb[8], lv[8], db[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int i = (i0 - q >= 1) ? i0 - q : 1;
                Lb[q] = Lc[((i - 1) << CS) + cc];
                lv[q] = lupt[((Tn - 1 - i) << CS) + cc];
                db[q] = sbeta[i - 1] - sbeta[i];                         // tempering.py:518-522
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int i = i0 - q;
                if (i >= 1) {
                    const double pacc = db[q] * (cL - Lb[q]);            // tempering.py:538
                    const bool sw = pacc > lv[q];                        // tempering.py:541
                    m |= sw ? (1u << (i & 31)) : 0u;
                    cL = sw ? cL : Lb[q];
                    if ((i & 31) == 0 || i == 1) {
                        smask[cc * MW + (i >> 5)] = m;
                        m = 0;
                    }
                }
            }
        }
    };
    // (the accepted rows - phase E - go out in the walk's shadow from every wave but the walking one)
    const bool walking = wv == (NW > 1 ? 1 : 0);
    auto store_accepted = [&]() {
        if constexpr (PIPE) {
            // system scope where a peer may pull the row; guests go home accepted or not.  Values and addresses of all passes
            // first, then the (inline asm) stores back to back: see phase E of k_stretch_fast.
            double2 val[NPASS];
            double* dstp[NPASS];
            bool on[NPASS];
#pragma unroll
            for (int p = 0; p < NPASS; ++p) {
                const int r = p * RPP + rsub;
                on[p] = false;
                val[p] = double2{0.0, 0.0};
                dstp[p] = A.pool;
                if (!rv[p]) continue;
                const int fl = s_flag[r];
                const double2 ql = CEN ? qkeep[p] : *reinterpret_cast<const double2*>(qtile + r * RS + jl * 2);
                const bool acc = (fl & 2) != 0;
                // (D = 128: a rejected guest's old row is read again here - a handful of rows per launch - instead of every row's
                //  registers staying live from the gathers to this point: 32 VGPRs, the second workgroup per CU.  Narrower rows
                //  keep the registers: at D = 64 the reload made the launch 1 us slower, 27.5 against 26.5)
                double2 o = sreg[p];
                if constexpr (DT == 128) {
                    o = double2{0.0, 0.0};
                    if (!acc && (fl & 8)) o = *reinterpret_cast<const double2*>(A.pool + row_off(s_rs[r], D, A.guest_delta) + jl * 2);
                }
                val[p].x = acc ? ql.x : o.x;
                val[p].y = acc ? ql.y : o.y;
                on[p] = (fl & (2 | 8)) != 0;
                dstp[p] = A.pool + (size_t)s_dst[r] * D + jl * 2;
            }
            if (A.sys_rows) {
#pragma unroll
                for (int p = 0; p < NPASS; ++p) if (on[p]) store_row16_sys(dstp[p], val[p]);
            } else {
#pragma unroll
                for (int p = 0; p < NPASS; ++p) if (on[p]) store_row16(dstp[p], val[p]);
            }
        } else {
#pragma unroll
            for (int p = 0; p < NPASS; ++p) {
                const int r = p * RPP + rsub;
                if (!rv[p]) continue;
                if ((s_flag[r] & 2) == 0) continue;
                store_row16(A.pool + (size_t)s_rs[r] * D + jl * 2,
                            CEN ? qkeep[p] : *reinterpret_cast<const double2*>(qtile + r * RS + jl * 2));
            }
        }
    };
    if (PIPE) {
        // The rank's hottest rung after the move goes to the hot neighbour - (L, P, row) of the 128 / Tl slots this
        // workgroup's columns meet there, slot order - BEFORE the wait for that neighbour's columns: its bottom boundary for
        // these very columns needs nothing else from this rank.  The rows themselves must be complete first (phase E, system
        // scope), so on a pipeline rank phase E does not hide in the walk's shadow.
        // (Round 5, measured and dropped: only the hottest rung's rows in front of the hand-off, the others in the walk's shadow as on
        //  one GPU - nothing at D = 32 / 64, +2 us at D = 128: profiles/r05c_pipe_overlap.txt)
        store_accepted();
        // (a rank's hand-off arguments are read from the kernarg segment HERE, late_kernarg: as by-value arguments they lived in SGPRs
        //  from the kernel's entry across the likelihood - the D = 32 instantiation spilled 250 of them into vector lanes, round 5)
        char* const box_l = late_kernarg<char*>(offsetof(FusedArgs, box));
        char* const box_hot_l = late_kernarg<char*>(offsetof(FusedArgs, box_hot));
        const int32_t par_l = late_kernarg<int32_t>(offsetof(FusedArgs, par));
        const uint32_t sweep_l = late_kernarg<uint32_t>(offsetof(FusedArgs, sweep));
        const long long budget_l = late_kernarg<long long>(offsetof(FusedArgs, budget));
        unsigned long long* const stats_l = late_kernarg<unsigned long long*>(offsetof(FusedArgs, stats));
        const PipeBox me = pipe_box(box_l, TG, W, D);
        if (has_top) {
            if (tid >= ((T - 1) << CS) && tid < (T << CS)) {
                const PipeBox hot = pipe_box(box_hot_l, TG, W, D);
                // (column-ordered records: the hot neighbour's column c meets the walker at index c - no permutation on its side)
                const int slot = COL ? c0 + (tid & (CB - 1)) : scol[tid];
                sys_store(hot.lp_dn + (size_t)(par_l * 2) * W + slot, Lc[tid]);
                sys_store(hot.lp_dn + (size_t)(par_l * 2 + 1) * W + slot, Pc[tid]);
                __hip_atomic_store(hot.ldn_loc + (size_t)par_l * W + slot, locc[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) pipe_raise(pipe_box(box_hot_l, TG, W, D).blk_ldn + blockIdx.x, sweep_l + 1);
            // what the hot neighbour's columns carry on leaving its rungs: the walk's top row
            if (walking) {
                if (lane == 0) pipe_spin(me.blk_lup + blockIdx.x, sweep_l + 1, budget_l, A.flags, stats_l ? stats_l + 4 : nullptr);
                if (lane < CB) {
                    const int c = c0 + lane, et = (T << CS) + lane;
                    Lc[et] = sys_load(me.lp_up + (size_t)(par_l * 2) * W + c);
                    Pc[et] = sys_load(me.lp_up + (size_t)(par_l * 2 + 1) * W + c);
                    locc[et] = pipe_guest_loc(par_l, 0, W, c);
                }
            }
        }
        if (walking && lane < CB) {
            // (straight-line walks for the shard heights of the BASELINE configs and their neighbours: Tl or Tl + 1 rungs.  Round 5: a
            //  16-rung rank walked through the run-time loop - 4 400 cycles for this phase against 1 540 on one GPU)
            if (TE == 9) walk(std::integral_constant<int, 9>{});
            else if (TE == 8) walk(std::integral_constant<int, 8>{});
            else if (TE == 16) walk(std::integral_constant<int, 16>{});
            else if (TE == 17) walk(std::integral_constant<int, 17>{});
            else if (TE == 4) walk(std::integral_constant<int, 4>{});
            else if (TE == 5) walk(std::integral_constant<int, 5>{});
            else if (TE == 1) smask[lane * MW] = 0;
            else walk(std::integral_constant<int, 0>{});
        }
    } else {
        if (!walking) store_accepted();
        if (walking && lane < CB) {
            if (T == 16) walk(std::integral_constant<int, 16>{});
            else if (T == 8) walk(std::integral_constant<int, 8>{});
            else if (T == 32) walk(std::integral_constant<int, 32>{});
            else walk(std::integral_constant<int, 0>{});
        }
    }
    lds_barrier();
    FUSED_TRACE(6);
#ifdef HENS_DEV_BUILD
    if (HENS_CUT_F == 5) return;
#endif

    // ---- phase G: permuted L / P / loc of the 128 slots, swap counts ----------------------------------------------
    auto bit = [&](int cc, int i) -> bool { return (i >= 1 && i < TE) && ((smask[cc * MW + (i >> 5)] >> (i & 31)) & 1u); };
    int se_n = 0;                                                        // the element that settles in this thread's slot
    auto settles = [&](const int e) -> int {                             // element that ends up in slot e: (rung it comes from, column)
        const int t = e >> CS, cc = e & (CB - 1);
        int st;
        if (MW == 1) {                                            // T <= 32: the whole column mask in one register
            const uint32_t mw = smask[cc];
            if ((mw >> t) & 1u) st = t - 1;                       // bit 0 is never set
            else st = t + __builtin_ctz(~(mw >> 1 >> t));         // consecutive swapped pairs directly above
        } else if (bit(cc, t)) {
            st = t - 1;
        } else {
            st = t;
            while (bit(cc, st + 1)) ++st;
        }
        return (st << CS) + cc;
    };
    // Written through where the launch goes out without a release fence, and on a pipeline rank (HIP stream, fences kept - cheaper
    // with nothing dirty: 8 x 16384 x 64 as a rank 52.3 -> 50.9 us, 16 x 4096 x 32 18.9 -> 18.4): TWO lanes per record - lane 2e writes
    // {L, P}, lane 2e + 1 {row, accept counter, slot index, -} - so that a record leaves the wave as ONE 32-byte sector (16-byte halves
    // from one lane in two instructions are two partial-sector writes, each a read-modify-write at the memory side, and the wait for
    // their acknowledgements is the launch's tail: 3 940 -> 1 870 cycles, config 2 16.5 -> 15.9 us per iteration; LABNOTES 10.13).  The
    // slot's own counter and index - they do not move with a walker - come from s_own (phase D).  Plain stores in the MH mix's
    // stretch iterations (HIP stream): there the records' halves and neighbours merge in L2.
    const bool paired = PIPE || late_kernarg<int32_t>(offsetof(FusedArgs, norel)) != 0;
    if (PIPE && tid < NEr) {
        se_n = settles(tid);
        if (has_top && (se_n >> CS) == T)                                // the hot neighbour's walker settles here, in a guest row:
            A.ghome[(size_t)(late_kernarg<int32_t>(offsetof(FusedArgs, par)) * 2) * W + c0 + (tid & (CB - 1))] = locc[((T - 1) << CS) + (tid & (CB - 1))];   // its home = the row of the walker that went up
    }
    if (paired ? tid < 2 * NEr : tid < NEr) {
        const int e = paired ? tid >> 1 : tid, t = e >> CS;
        const int se = settles(e);
        const size_t di = (size_t)t * W + scol[e];
        WalkerRec* const wrecnew_l = late_kernarg<WalkerRec*>(offsetof(FusedArgs, wrecnew));
        int32_t* const locnew_l = late_kernarg<int32_t*>(offsetof(FusedArgs, locnew));
        if (paired) {
            const int h = tid & 1;
            const int2 own = s_own[e];
            const double2 v = h == 0 ? double2{Lc[se], Pc[se]} : double2{__hiloint2double(own.x, locc[se]), __hiloint2double(0, own.y)};
            store_row16(reinterpret_cast<double*>(&wrecnew_l[di]) + 2 * h, v);
            if (h == 1) wt_store(&locnew_l[di], locc[se]);
        } else {
            wrecnew_l[di] = make_wrec(Lc[se], Pc[se], locc[se], wr_n.acc, wr_n.slot);
            locnew_l[di] = locc[se];
        }
    }
    for (int i = 1 + tid; i < TE; i += NT) {                             // pair (i, i-1) -> index i-1
        unsigned n = 0;
        for (int cc = 0; cc < CB; ++cc) n += bit(cc, i) ? 1u : 0u;
        // (a pipeline rank too - round 3: one ticket per workgroup on ONE address, for a collector at the end of the launch, cost
        //  16 ns per workgroup, serialised: 17 us at 1024 workgroups; the next launch sums these rows and publishes the counts)
        if (n) {
            uint32_t* const acc_l = late_kernarg<uint32_t*>(offsetof(FusedArgs, swap_acc));
            const int32_t rows_l = late_kernarg<int32_t>(offsetof(FusedArgs, acc_rows));
            atomicAdd(&acc_l[(size_t)(blockIdx.x & (rows_l - 1)) * (TE - 1) + (i - 1)], n);
        }
    }
    // ---- phase E, the walking wave's share ---------------------------------------------------------------------------
    if (!PIPE && walking) store_accepted();

    if constexpr (PIPE) {
        char* const box_l = late_kernarg<char*>(offsetof(FusedArgs, box));
        char* const box_cold_l = late_kernarg<char*>(offsetof(FusedArgs, box_cold));
        const double* const pool_cold_l = late_kernarg<const double*>(offsetof(FusedArgs, pool_cold));
        const int32_t par_l = late_kernarg<int32_t>(offsetof(FusedArgs, par));
        const uint32_t sweep_l = late_kernarg<uint32_t>(offsetof(FusedArgs, sweep));
        const long long budget_l = late_kernarg<long long>(offsetof(FusedArgs, budget));
        unsigned long long* const stats_l = late_kernarg<unsigned long long*>(offsetof(FusedArgs, stats));
        const PipeBox me = pipe_box(box_l, TG, W, D);
        // ---- my columns leave for the cold neighbour: its workgroup with the same index may start its walk -----------------
        if (has_bot) {
            const PipeBox cold = pipe_box(box_cold_l, TG, W, D);
            if (tid < CB) {                                              // (slot threads of my coldest rung: t = 0, cc = tid)
                sys_store(cold.lp_up + (size_t)(par_l * 2) * W + c0 + tid, Lc[se_n]);
                sys_store(cold.lp_up + (size_t)(par_l * 2 + 1) * W + c0 + tid, Pc[se_n]);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) pipe_raise(cold.blk_lup + blockIdx.x, sweep_l + 1);
            // ---- bottom boundary, hot side: pair (g, g-1) for the same columns (needs the cold neighbour's rung after ITS move,
            // not its walk); a walker may fall through all my rungs in one sweep, so the rows from above must have landed too
            if (tid == 0) pipe_spin(me.blk_ldn + blockIdx.x, sweep_l + 1, budget_l, A.flags, stats_l ? stats_l + 6 : nullptr);
            if (tid == 64 && has_top) pipe_spin(me.blk_rows + blockIdx.x, sweep_l + 1, budget_l, A.flags, stats_l ? stats_l + 8 : nullptr);
            __syncthreads();
            if (tid < CB) {
                const int cc = tid, c = c0 + cc, g = R0;
                int slot_below = c;                                                  // (column-ordered records: published by column)
                if (!COL) {
                    const uint4* kp = reinterpret_cast<const uint4*>(A.keys) + (size_t)(g - 1) * 2;
                    const uint4 ka = kp[0], kb = kp[1];
                    const uint32_t key[8] = {ka.x, ka.y, ka.z, ka.w, kb.x, kb.y, kb.z, kb.w};
                    slot_below = (int)prp((uint32_t)c, key, A.idx_bits, (uint32_t)W);
                }
                const double La = Lc[se_n];
                const double Lb = sys_load(me.lp_dn + (size_t)(par_l * 2) * W + slot_below);
                const double db = A.betas[g - 1] - A.betas[g];                       // tempering.py:518-522
                int32_t src = PIPE_NOSEL, yrow = 0;
                if (db * (La - Lb) > log(pt_uniform(A.seed, A.iter, TG - 1 - g, W, c))) {   // tempering.py:535-541
                    src = locc[se_n];
                    yrow = __hip_atomic_load(me.ldn_loc + (size_t)par_l * W + slot_below, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    const double Pb = sys_load(me.lp_dn + (size_t)(par_l * 2 + 1) * W + slot_below);
                    const int32_t gl = pipe_guest_loc(par_l, 1, W, c);
                    const size_t di = (size_t)scol[tid];                             // (rung 0 of the rank)
                    late_kernarg<WalkerRec*>(offsetof(FusedArgs, wrecnew))[di] = make_wrec(Lb, Pb, gl, wr_n.acc, wr_n.slot);
                    late_kernarg<int32_t*>(offsetof(FusedArgs, locnew))[di] = gl;
                    // its home: the row of the walker that goes down - or, if that one only just fell in from above (a guest
                    // itself), the row of the walker that went up across the top boundary
                    A.ghome[(size_t)(par_l * 2 + 1) * W + c] = src >= 0 ? src : locc[((T - 1) << CS) + cc];
                }
                s_src[cc] = src;
                s_yrow[cc] = yrow;
            }
            __syncthreads();
            {
                double* dst = cold.guest + (size_t)(par_l * 2) * W * D;              // rows that move down: push
                double* mine = me.guest + (size_t)(par_l * 2 + 1) * W * D;           // rows that move up: pull
                // all of a thread's loads first - the pulls cross xGMI, their latencies must overlap - then the stores
                for (int base = 0; base < CB * D; base += 4 * NT) {
                    double push[4], pull[4];
                    bool on[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int idx = base + q * NT + tid;
                        on[q] = false;
                        if (idx < CB * D) {
                            const int col = idx / D, d = idx - col * D;
                            const int32_t src = s_src[col];
                            if (src != PIPE_NOSEL) {
                                on[q] = true;
                                push[q] = sys_load(A.pool + row_off(src, D, A.guest_delta) + d);
                                pull[q] = sys_load(pool_cold_l + (size_t)s_yrow[col] * D + d);
                            }
                        }
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int idx = base + q * NT + tid;
                        if (on[q]) {
                            const int col = idx / D, d = idx - col * D;
                            sys_store(dst + (size_t)(c0 + col) * D + d, push[q]);
                            mine[(size_t)(c0 + col) * D + d] = pull[q];
                        }
                    }
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) pipe_raise(cold.blk_rows + blockIdx.x, sweep_l + 1);
            // (that ALL rows from above have landed - what the cold neighbour's next iteration waits for - is said by this
            //  rank's NEXT launch: StretchArgs::rt_flag, k_pipe_epilogue; no grid-wide ticket here)
        }
    }
    if constexpr (!PIPE) if (late_kernarg<int32_t>(offsetof(FusedArgs, norel))) launch_end_wait();        // (no release fence on this launch's packet - see wt_store)
    FUSED_TRACE(7);
#undef FUSED_TRACE
}


// Stand-alone ladder adaptation (one workgroup): used where it cannot ride in the next stretch
// launch (parity API, sharded ladder, generic row widths, T > 64, end of a hens_step call).
// Lane-parallel where the reference's arithmetic allows: ratios, dS, deltaT per lane; the cumsum
// stays a sequential left-to-right sum like np.cumsum; reciprocal and update per lane again.
inline __global__ __launch_bounds__(256) void k_adapt(const AdaptArgs A) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    constexpr int NTHREADS = 256;
    const int T = A.T, tid = threadIdx.x;
    double* r = reinterpret_cast<double*>(smem_raw);
    double* dT = r + T;
    double* bnew = dT + T;
    unsigned* cnt = reinterpret_cast<unsigned*>(bnew + T);
    for (int j = tid; j < T; j += NTHREADS) cnt[j] = 0;
    if (A.wait_flags && tid < A.wait_n) pipe_spin(A.wait_flags + tid, A.wait_target, A.wait_budget, A.wait_err);
    __syncthreads();
    const int total = A.nblocks * (T - 1);
    for (int e0 = tid; e0 < total; e0 += 8 * NTHREADS) {
        unsigned v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int e = e0 + q * NTHREADS;
            v[q] = (e < total) ? A.swap_part[e] : 0u;
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int e = e0 + q * NTHREADS;
            if (v[q]) {
                atomicAdd(&cnt[e % (T - 1)], v[q]);
                if (A.zero_after) A.swap_part[e] = 0u;
            }
        }
    }
    __syncthreads();
    for (int j = tid; j < T; j += NTHREADS) {
        bnew[j] = A.betas_in[j];
        if (j < T - 1) r[j] = (double)cnt[j] / (double)A.W;           // :587
    }
    __syncthreads();
    if (A.moving) {
        const double decay = A.lag / ((double)A.time + A.lag);        // :571
        const double kappa = decay / A.nu;                            // :572
        for (int j = tid; j + 2 < T; j += NTHREADS) {
            const double dS = kappa * (r[j] - r[j + 1]);              // :575
            double d = 1.0 / bnew[j + 1] - 1.0 / bnew[j];             // :578
            d *= exp(dS);
            dT[j] = d;
        }
        __syncthreads();
        if (tid == 0)
            for (int j = 1; j + 2 < T; ++j) dT[j] = dT[j - 1] + dT[j];    // np.cumsum order
        __syncthreads();
        const double inv0 = 1.0 / bnew[0];
        for (int j0 = 0; j0 + 2 < T; j0 += NTHREADS) {
            const int j = j0 + tid;
            double upd = 0.0;
            if (j + 2 < T) {
                const double bn = 1.0 / (dT[j] + inv0);               // :580
                upd = bnew[j + 1] + (bn - bnew[j + 1]);               // :583,:593
            }
            __syncthreads();
            if (j + 2 < T) bnew[j + 1] = upd;
        }
        __syncthreads();
    }
    if (A.zero_rows)
        for (int e = tid; e < total; e += NTHREADS) A.zero_rows[e] = 0u;
    for (int j = tid; j < T; j += NTHREADS) {
        A.betas_out[j] = bnew[j];
        if (j < T - 1) {
            A.swaps_last[j] = (double)cnt[j];
            A.swaps_total[j] += (double)cnt[j];
        }
    }
}

// debug (hens_debug_draws): the Philox-mode PT draws of iteration `it` in the form the cascade consumes them -
// slot[t][c] = slot of rung t that column c visits, u[j][c] = the swap uniform of pair (T-1-j, T-2-j) on column c
inline __global__ void k_debug_pt(int32_t* slot, double* u, int T, int W, int idx_bits, uint64_t seed, uint64_t it) {
    const int64_t n = (int64_t)T * W;
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
        const int t = (int)(e / W), c = (int)(e - (int64_t)t * W);
        slot[e] = pt_slot(seed, it, t, T, c, idx_bits, W);
        if (t < T - 1) u[e] = pt_uniform(seed, it, t, W, c);
    }
}

// debug: materialise one PRP permutation (tests/test_hip_parity.py checks bijectivity and uniformity)
inline __global__ void k_debug_prp(int32_t* out, int W, int idx_bits, uint64_t seed, uint64_t it, uint32_t purpose, uint32_t rung) {
    const PrpKey K = prp_key(seed, it, purpose, rung);
    for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < W; c += gridDim.x * blockDim.x)
        out[c] = (int32_t)prp((uint32_t)c, K.k, idx_bits, (uint32_t)W);
}

// ---------------------------------------------------------------------------------------------
// Host-callable likelihood: proposal and accept/update stay on the device, the caller
// evaluates log_like_fn on the proposed points in between (ensemble.py:1219-1545 contract).
// Throughput here is set by the host function and two PCIe hops per half-step, not by these kernels.
// ---------------------------------------------------------------------------------------------
struct HostLikeArgs {
    double* pool;
    int32_t* loc;
    double* L;
    double* P;
    const double* betas;
    Draws dr;
    const double* lo;
    const double* hi;
    const double* period;    // [D] or nullptr (see StretchArgs::period)
    double* qbuf;            // [Tl][Ns][D] proposed points
    uint8_t* inbox;          // [Tl][Ns] 1 = inside the prior box
    int32_t* rs_old;         // [Tl][Ns] pool row of the moving walker before the update
    uint8_t* keep;           // [Tl][Ns]
    const double* logl;      // [Tl][Ns] from the caller
    const double* u_acc;     // [Tl][Ns]
    uint32_t* accepted;
    unsigned* flags;
    double logp_in;
    int32_t Tl, W, D, split, N0, rung_begin, home_off, tempered;
    int32_t ns_x, soff_x;    // see StretchArgs::ns_x
};

inline __global__ void k_propose(const HostLikeArgs A) {
    const int Ns = A.ns_x ? A.ns_x : (A.split == 0 ? A.N0 : A.W - A.N0), s_off = A.ns_x ? A.soff_x : (A.split == 0 ? 0 : A.N0);
    const int64_t total = (int64_t)A.Tl * Ns * A.D;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t wk = i / A.D;
        const int d = (int)(i - wk * A.D);
        const int tl = (int)(wk / Ns), k = (int)(wk - (int64_t)tl * Ns);
        const size_t di = (size_t)tl * A.W + s_off + k;
        const int own = A.dr.own[di], cw = A.dr.cw[di];
        const double zz = A.dr.zz[di];
        const double sv = A.pool[(size_t)A.loc[tl * A.W + own] * A.D + d];
        const double cv = A.pool[(size_t)A.loc[tl * A.W + cw] * A.D + d];
        double qv = cv - (cv - sv) * zz;                             // stretch.py:143,145
        if (A.period) qv = periodic_wrap(cv - periodic_diff(sv, cv, A.period[d]) * zz, A.period[d]);   // stretch.py:136-154
        A.qbuf[i] = qv;
        if (!((qv >= A.lo[d]) && (qv <= A.hi[d]))) A.inbox[wk] = 0;  // prior.py:80-88 (inbox preset to 1)
        if (!(fabs(qv) < INFINITY)) atomicOr(A.flags, FLAG_NONFINITE_X);
    }
}

inline __global__ void k_accept_decide(const HostLikeArgs A) {
    const int Ns = A.ns_x ? A.ns_x : (A.split == 0 ? A.N0 : A.W - A.N0), s_off = A.ns_x ? A.soff_x : (A.split == 0 ? 0 : A.N0);
    const int64_t total = (int64_t)A.Tl * Ns;
    for (int64_t wk = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; wk < total; wk += (int64_t)gridDim.x * blockDim.x) {
        const int tl = (int)(wk / Ns), k = (int)(wk - (int64_t)tl * Ns);
        const size_t di = (size_t)tl * A.W + s_off + k;
        const int own = A.dr.own[di];
        const size_t gi = (size_t)tl * A.W + own;
        const bool inbox = A.inbox[wk] != 0;
        double logl = A.logl[wk];
        if (logl != logl) {                                            // red_blue.py:279-281
            logl = -1e300;
            atomicOr(A.flags, FLAG_NAN_LOGL);
        }
        const double logp = inbox ? A.logp_in : -INFINITY;
        const double Lold = A.L[gi], Pold = A.P[gi];
        double logP, prevP;
        if (A.tempered) {                                              // tempering.py:304-306,343-349
            const double beta = A.betas[A.rung_begin + tl];
            double lt = logl * beta;
            if (lt != lt) lt = -INFINITY;
            logP = lt + logp;
            double lo_ = Lold * beta;
            if (lo_ != lo_) lo_ = -INFINITY;
            prevP = lo_ + Pold;
        } else {
            logP = logl + logp;
            prevP = Lold + Pold;
        }
        const double lnpdiff = A.dr.fac[di] + logP - prevP;           // red_blue.py:292
        const bool keep = lnpdiff > log(A.u_acc[wk]);                  // red_blue.py:294
        if (keep) {                                                    // move.py:513-532
            A.L[gi] = logl;
            A.P[gi] = (fabs(logp) == INFINITY) ? 0.0 : logp;
            atomicAdd(&A.accepted[gi], 1u);
        }
        A.rs_old[wk] = A.loc[gi];
        A.loc[gi] = A.home_off + tl * A.W + own;
        A.keep[wk] = keep ? 1 : 0;
    }
}

inline __global__ void k_accept_rows(const HostLikeArgs A) {
    const int Ns = A.ns_x ? A.ns_x : (A.split == 0 ? A.N0 : A.W - A.N0), s_off = A.ns_x ? A.soff_x : (A.split == 0 ? 0 : A.N0);
    const int64_t total = (int64_t)A.Tl * Ns * A.D;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t wk = i / A.D;
        const int d = (int)(i - wk * A.D);
        const int tl = (int)(wk / Ns), k = (int)(wk - (int64_t)tl * Ns);
        const int own = A.dr.own[(size_t)tl * A.W + s_off + k];
        const double v = A.keep[wk] ? A.qbuf[i] : A.pool[(size_t)A.rs_old[wk] * A.D + d];
        A.pool[(size_t)(A.home_off + tl * A.W + own) * A.D + d] = v;
    }
}

// column-order decisions -> the reference's k order (row j, element colk[j][c])
inline __global__ void k_pt_sel_to_korder(const uint8_t* __restrict__ selcol, const int32_t* __restrict__ colk,
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
inline __global__ void k_xchg_count(const int32_t* __restrict__ srcfull, const int32_t* __restrict__ rank_of_rung,
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
inline __global__ void k_xchg_fill(const int32_t* __restrict__ srcfull, const int32_t* __restrict__ rank_of_rung,
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

inline __global__ void k_pack_rows(const double* __restrict__ pool, const int32_t* __restrict__ loc,
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

inline __global__ void k_unpack_rows(double* __restrict__ pool, int32_t* __restrict__ locnew, double* __restrict__ Pnew,
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

// ---------------------------------------------------------------------------------------------
// Ladder pipeline: the sharded ladder's PT sweep as a neighbour exchange of one-sided puts.
//
// Rank r keeps the global rungs [b, e) (rung 0 = beta 1 is the coldest).  The cascade runs hot ->
// cold (tempering.py:515-541), so the only true dependency between ranks is "the column state at the
// top of my rungs", which the hot neighbour knows when ITS walk is done.  Per sweep and boundary
// (hot rung e on rank r+1, cold rung e-1 on rank r):
//   r   -> r+1 : (L, P) of rung e-1 after the stretch move, slot order, + where its rows are
//                                                       [stretch kernel, phase D; k_pipe_pub for generic D]
//   r+1 -> r   : (L, P) of the walker each column carries on leaving rung e       [k_pipe_walk]
//   both sides evaluate the boundary pair from identical Philox draws; then the HOT side moves the rows
//   in both directions, so that its next iteration never waits for the (lagging) cold side's walk:
//   r+1 -> r   : PUSH the rows that move down (sparse, at their column index)     [k_pipe_bottom]
//   r+1 <- r   : PULL the rows that move up straight out of rank r's pool         [k_pipe_bottom]
//   every rank -> all ranks: swap counts of the pairs it owns (ladder adaptation) [k_pipe_walk, last workgroup]
// Every message is a store straight into the peer's MAILBOX (uncached device memory, peer-mapped
// through HIP IPC: xGMI stores on a multi-GPU node) followed by a flag raised by the last workgroup of
// the producing launch; the consuming kernel spins on the flag in its prologue.  No host round trip,
// no packing, no counts, no extra launches: a row that moves lands in a guest area at its COLUMN index,
// and `loc` simply points there (row_off) until the next stretch move rewrites every walker into its
// home row anyway.  Mailbox buffers are double-buffered by sweep parity; the flag protocol itself keeps
// a rank at most one sweep ahead of its neighbours.
// ---------------------------------------------------------------------------------------------
constexpr int PIPE_MAX_RANKS = PIPE_FLAG_WORDS - PF_CNT0;

struct PipeArgs {
    const double* pool;
    int64_t guest_delta;
    const double* L; const double* P; const int32_t* loc;     // current [Tl][W] (all rows at home after the stretch move)
    double* Lnew; double* Pnew; int32_t* locnew;              // next
    const double* betas;          // [T] whole ladder (replicated)
    char* box;                    // my mailbox
    char* box_hot;                // hot neighbour's (rank + 1) or nullptr
    char* box_cold;               // cold neighbour's (rank - 1) or nullptr
    const double* pool_cold;      // cold neighbour's walker pool (rows that move up are read from it)
    char* const* boxes;           // [nranks] every mailbox (swap counts)
    double* Lcur; double* Pcur; int32_t* botsrc;              // [W] what each column carries below my coldest rung
    uint32_t* swap_part;          // [nblocks][TE-1]
    unsigned* flags;              // context error flags
    unsigned* tickets;            // [4] workgroup tickets: 0 walk, 1 bottom, 2 stretch publish
    unsigned long long* stats;    // debug wait statistics [16] (pairs: ticks, waits) or nullptr
    uint64_t iter, seed;
    uint32_t sweep;               // pipeline sweep counter (flags carry sweep + 1)
    long long budget;             // wall-clock ticks a flag wait may take
    int32_t T, W, D, Tl, rung_begin, idx_bits, par, nranks, rank;
    int32_t home_off;             // pool row of (rung 0, slot 0) after this iteration's stretch move
    int32_t nowait;               // staged (RCCL) transport: messages arrive in stream order, nothing to spin on
    int32_t fuse_bottom;          // the walk kernel runs the bottom boundary as its own last phase (one-sided transport)
    int32_t count_tail;           // 1: workgroup 0 of the walk reduces and publishes the swap counts; 0: the adapting
                                  // workgroup of the next iteration's first launch does (StretchArgs::cnt_push)
};

__device__ __forceinline__ int pipe_slot(const PipeArgs& A, int g, int c) {
    return pt_slot(A.seed, A.iter, g, A.T, c, A.idx_bits, A.W);
}
// log-uniform of pair (i, i-1), column c - keyed exactly like k_pt_cascade (row T-1-i)
__device__ __forceinline__ double pipe_logu(const PipeArgs& A, int i, int c) {
    return log(pt_uniform(A.seed, A.iter, A.T - 1 - i, A.W, c));     // tempering.py:535
}

struct PipeWaitArgs {
    const unsigned* p[4];
    const unsigned* cnt_flags;    // my flags + PF_CNT0 (or nullptr): wait for every rank's counts
    unsigned* err;
    long long budget;             // wall_clock64 ticks
    uint32_t target;
    int32_t n, nranks;
};
inline __global__ void k_pipe_wait(const PipeWaitArgs A) {
    const int i = threadIdx.x;
    const unsigned* f = nullptr;
    if (i < A.n) f = A.p[i];
    else if (A.cnt_flags && i - A.n < A.nranks) f = A.cnt_flags + (i - A.n);
    if (f) pipe_spin(f, A.target, A.budget, A.err);
}
// End of a hens_step call on a rank that steps with the fused iteration: what the head of the NEXT iteration's first launch
// would say - the previous sweep's pushes are complete (cold neighbour's PF_ROWS_TOP), and, on the reference's adaptation
// schedule, the last sweep's swap counts (the adaptation that closes the call needs every rank's).
struct PipeEpilogueArgs {
    unsigned* rt_flag; uint32_t rt_value;
    const uint32_t* cp_rows; char* const* cp_boxes;
    uint32_t cp_sweep;
    int32_t push, cp_nblocks, cp_np, cp_nranks, cp_rank, cp_T, rung_begin, W, D;
};
inline __global__ void k_pipe_epilogue(const PipeEpilogueArgs A) {
    if (A.rt_flag && threadIdx.x == 0) pipe_raise(A.rt_flag, A.rt_value);
    if (A.push) pipe_push_counts(A.cp_rows, A.cp_nblocks, A.cp_np, A.cp_boxes, A.cp_nranks, A.cp_rank, A.cp_T, A.rung_begin, A.W, A.D,
                                 A.cp_sweep, (int)threadIdx.x);
}
// my hottest rung after the stretch move -> hot neighbour, flag included (one workgroup).  Only for row widths
// without a fast stretch kernel: those publish from their own accept phase (StretchArgs::pub_lp).
inline __global__ __launch_bounds__(1024) void k_pipe_pub(const PipeArgs A) {
    const PipeBox hot = pipe_box(A.box_hot, A.T, A.W, A.D);
    const int W = A.W;
    const size_t base = (size_t)(A.Tl - 1) * W;
    for (int i = threadIdx.x; i < 2 * W; i += blockDim.x) {
        const int w = i < W ? i : i - W;
        sys_store(hot.lp_dn + (size_t)(A.par * 2 + (i < W ? 0 : 1)) * W + w, i < W ? A.L[base + w] : A.P[base + w]);
    }
    if (threadIdx.x == 0)
        __hip_atomic_store(hot.meta + A.par, (long long)A.home_off + (long long)(A.Tl - 1) * W, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) pipe_raise(hot.flags + PF_LDN, A.sweep + 1);
}

constexpr int PIPE_COLS = 16;       // columns per workgroup of the bottom-boundary kernel (W / 16 workgroups: every CU busy)

// bottom boundary, hot side: decide pair (b, b-1), settle my coldest rung, push the rows that move down into
// the cold neighbour's guest area and pull the rows that move up out of its pool.  One workgroup = PIPE_COLS
// columns starting at c0.  sh_La / sh_bsrc: what each of the columns carries below my coldest rung, in LDS (when
// the walk kernel runs this as its own last phase) or nullptr (stand-alone kernel: from A.Lcur / A.botsrc).
__device__ __forceinline__ void pipe_bottom_block(const PipeArgs& A, const int c0, int32_t* s_src, int32_t* s_below,
                                                  const double* sh_La, const int32_t* sh_bsrc) {
    const int W = A.W, D = A.D;
    const PipeBox me = pipe_box(A.box, A.T, W, D);
    const bool has_top = A.rung_begin + A.Tl < A.T;
    // the cold neighbour's rung after ITS stretch move; a walker may fall through all my rungs in one sweep,
    // so the rows from above must have landed too
    if (threadIdx.x == 0 && !A.nowait) pipe_spin(me.flags + PF_LDN, A.sweep + 1, A.budget, A.flags, A.stats ? A.stats + 6 : nullptr);
    if (threadIdx.x == 64 && has_top && !A.nowait) pipe_spin(me.flags + PF_ROWS_TOP, A.sweep + 1, A.budget, A.flags, A.stats ? A.stats + 8 : nullptr);
    __syncthreads();
    if (threadIdx.x < PIPE_COLS) {
        const int c = c0 + threadIdx.x;
        int32_t src = PIPE_NOSEL, sb = 0;
        if (c < W) {
            const int g = A.rung_begin;                              // my coldest rung; the pair is (g, g-1)
            const int slot = pipe_slot(A, g, c), slot_below = pipe_slot(A, g - 1, c);
            sb = slot_below;
            const double La = sh_La ? sh_La[threadIdx.x] : A.Lcur[c];
            const double Lb = sys_load(me.lp_dn + (size_t)(A.par * 2) * W + slot_below);
            const double db = A.betas[g - 1] - A.betas[g];
            if (db * (La - Lb) > pipe_logu(A, g, c)) {
                src = sh_bsrc ? sh_bsrc[threadIdx.x] : A.botsrc[c];
                A.Lnew[slot] = Lb;
                A.Pnew[slot] = sys_load(me.lp_dn + (size_t)(A.par * 2 + 1) * W + slot_below);
                A.locnew[slot] = pipe_guest_loc(A.par, 1, W, c);
            }
        }
        s_src[threadIdx.x] = src;
        s_below[threadIdx.x] = sb;
    }
    __syncthreads();
    const PipeBox cold = pipe_box(A.box_cold, A.T, W, D);
    double* dst = cold.guest + (size_t)(A.par * 2) * W * D;                     // rows that move down: push
    double* mine = me.guest + (size_t)(A.par * 2 + 1) * W * D;                  // rows that move up: pull
    const long long cold_home = __hip_atomic_load(me.meta + A.par, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    // all of a thread's loads first - the pulls cross xGMI, their latencies must overlap - then the stores
    for (int base = 0; base < PIPE_COLS * D; base += 8 * (int)blockDim.x) {
        double push[8], pull[8];
        bool on[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int idx = base + q * (int)blockDim.x + (int)threadIdx.x;
            on[q] = false;
            if (idx < PIPE_COLS * D) {
                const int col = idx / D, d = idx - col * D;
                const int32_t src = s_src[col];
                if (src != PIPE_NOSEL) {
                    on[q] = true;
                    push[q] = A.pool[row_off(src, D, A.guest_delta) + d];
                    pull[q] = sys_load(A.pool_cold + (size_t)(cold_home + s_below[col]) * D + d);
                }
            }
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int idx = base + q * (int)blockDim.x + (int)threadIdx.x;
            if (on[q]) {
                const int col = idx / D, d = idx - col * D;
                sys_store(dst + (size_t)(c0 + col) * D + d, push[q]);
                mine[(size_t)(c0 + col) * D + d] = pull[q];
            }
        }
    }
    // the last workgroup to get here tells the cold neighbour that its rows from above have landed
    if (pipe_last_ticket(A.tickets + 1, gridDim.x * (A.sweep + 1u)) && threadIdx.x == 0) pipe_raise(cold.flags + PF_ROWS_TOP, A.sweep + 1);
}

// The walk over my rungs (+ the virtual rung of the hot neighbour on top): k_pt_cascade on the extended
// ladder; the pair across my top boundary is the first step of every column.
inline __global__ __launch_bounds__(PT_THREADS) void k_pipe_walk(const PipeArgs A) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int T = A.T, W = A.W, Tl = A.Tl, D = A.D;
    const bool has_top = A.rung_begin + Tl < T, has_bot = A.rung_begin > 0;
    const int TE = Tl + (has_top ? 1 : 0);
    const size_t NE = (size_t)TE * PT_COLS;
    double* Lc = reinterpret_cast<double*>(smem_raw);            // [TE][PT_COLS]
    double* lu = Lc + NE;                                        // [TE][PT_COLS] row i = pair (i, i-1), extended indices
    double* Pc = lu + NE;
    double* sbeta = Pc + NE;                                     // [TE]
    int32_t* locc = reinterpret_cast<int32_t*>(sbeta + TE);
    int32_t* scol = locc + NE;
    uint32_t* smask = reinterpret_cast<uint32_t*>(scol + NE);    // [PT_COLS][MW]
    const int MW = (TE + 31) / 32;
    const int tid = threadIdx.x;
    const int c0 = blockIdx.x * PT_COLS;
    const PipeBox me = pipe_box(A.box, T, W, D);
    __shared__ double sh_La[PT_COLS];                            // what each column carries below my coldest rung
    __shared__ int32_t sh_bsrc[PT_COLS], sh_src[PT_COLS], sh_below[PT_COLS];

    if (has_top && !A.nowait) {                                  // what the hot neighbour's columns carry must be here
        if (tid == 0) pipe_spin(me.lupf + blockIdx.x, A.sweep + 1, A.budget, A.flags, A.stats ? A.stats + 4 : nullptr);
        __syncthreads();
    }
    for (int t = tid; t < TE; t += PT_THREADS) sbeta[t] = A.betas[A.rung_begin + t];
    for (int e = tid; e < (int)NE; e += PT_THREADS) {
        const int t = e / PT_COLS, cc = e - t * PT_COLS, c = c0 + cc;
        if (c >= W) continue;
        const int g = A.rung_begin + t;
        if (t == Tl) {
            scol[e] = c;
            Lc[e] = sys_load(me.lp_up + (size_t)(A.par * 2) * W + c);
            Pc[e] = sys_load(me.lp_up + (size_t)(A.par * 2 + 1) * W + c);
            locc[e] = pipe_guest_loc(A.par, 0, W, c);
        } else {
            const int slot = pipe_slot(A, g, c);
            scol[e] = slot;
            Lc[e] = A.L[(size_t)t * W + slot];
            Pc[e] = A.P[(size_t)t * W + slot];
            locc[e] = A.loc[(size_t)t * W + slot];
        }
        if (t >= 1) lu[e] = pipe_logu(A, g, c);
    }
    __syncthreads();

    if (tid < PT_COLS) {
        if (c0 + tid < W) {
            const int cc = tid;
            double cL = Lc[(size_t)(TE - 1) * PT_COLS + cc];
            uint32_t m = 0;
            if (TE == 1) smask[cc * MW] = 0;
            for (int i0 = TE - 1; i0 >= 1; i0 -= 8) {                  // 8 steps per LDS round trip
                double Lb[8], lv[8], db[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int i = (i0 - q >= 1) ? i0 - q : 1;
                    Lb[q] = Lc[(size_t)(i - 1) * PT_COLS + cc];
                    lv[q] = lu[(size_t)i * PT_COLS + cc];
                    db[q] = sbeta[i - 1] - sbeta[i];                   // tempering.py:518-522
                }
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int i = i0 - q;
                    if (i >= 1) {
                        const bool sw = db[q] * (cL - Lb[q]) > lv[q];  // :538,:541
                        m |= sw ? (1u << (i & 31)) : 0u;
                        cL = sw ? cL : Lb[q];
                        if ((i & 31) == 0 || i == 1) {
                            smask[cc * MW + (i >> 5)] = m;
                            m = 0;
                        }
                    }
                }
            }
        }
    }
    __syncthreads();

    auto bit = [&](int cc, int i) -> bool { return (i >= 1 && i < TE) && ((smask[cc * MW + (i >> 5)] >> (i & 31)) & 1u); };
    PipeBox cold{};
    if (has_bot) cold = pipe_box(A.box_cold, T, W, D);
    for (int e = tid; e < Tl * PT_COLS; e += PT_THREADS) {
        const int t = e / PT_COLS, cc = e - t * PT_COLS, c = c0 + cc;
        if (c >= W) continue;
        int st;
        if (MW == 1) {                                            // TE <= 32: the whole column mask in one register
            const uint32_t mw = smask[cc];
            if ((mw >> t) & 1u) st = t - 1;                       // bit 0 is never set
            else st = t + __builtin_ctz(~(mw >> 1 >> t));         // consecutive swapped pairs directly above
        } else if (bit(cc, t)) {
            st = t - 1;
        } else {
            st = t;
            while (bit(cc, st + 1)) ++st;
        }
        const int se = st * PT_COLS + cc;
        const size_t di = (size_t)t * W + scol[e];
        A.Lnew[di] = Lc[se];
        A.Pnew[di] = Pc[se];
        A.locnew[di] = locc[se];
        if (t == 0) {                                            // the column leaves my rungs with this walker
            A.Lcur[c] = Lc[se];
            A.Pcur[c] = Pc[se];
            A.botsrc[c] = locc[se];
            sh_La[cc] = Lc[se];
            sh_bsrc[cc] = locc[se];
            if (has_bot) {
                sys_store(cold.lp_up + (size_t)(A.par * 2) * W + c, Lc[se]);
                sys_store(cold.lp_up + (size_t)(A.par * 2 + 1) * W + c, Pc[se]);
            }
        }
    }
    for (int i = 1 + tid; i < TE; i += PT_THREADS) {
        unsigned n = 0;
        for (int cc = 0; cc < PT_COLS && c0 + cc < W; ++cc) n += bit(cc, i) ? 1u : 0u;
        __hip_atomic_store(&A.swap_part[(size_t)blockIdx.x * (TE - 1) + (i - 1)], n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // my 16 columns are in the cold neighbour's mailbox: its walk's block with the same index may start
    if (has_bot) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) pipe_raise(cold.lupf + blockIdx.x, A.sweep + 1);
    }
    // the pair across my bottom boundary for the same 16 columns (needs the cold neighbour's rung, not its walk)
    static_assert(PIPE_COLS == PT_COLS, "the fused bottom phase works on the walk's columns");
    if (A.fuse_bottom && has_bot) pipe_bottom_block(A, c0, sh_src, sh_below, sh_La, sh_bsrc);
    if (!A.count_tail) return;

    // ---- workgroup 0 speaks for the launch once everyone has arrived: the swap counts of my pairs ------
    const long long dbg_t0 = A.stats ? wall_clock64() : 0;         // (debug statistics only: ~1.5 us per reading)
    if (!pipe_arrive_collect(A.tickets + 0, gridDim.x, A.sweep, A.budget, A.flags)) return;
    const long long dbg_t1 = A.stats ? wall_clock64() : 0;
    const int NP = TE - 1;
    unsigned* s_n = reinterpret_cast<unsigned*>(smem_raw);           // [NP] (the column tables are dead)
    for (int i = tid; i < NP; i += PT_THREADS) s_n[i] = 0;
    __syncthreads();
    const int total = (int)gridDim.x * NP;
    for (int e0 = tid; e0 < total; e0 += 16 * PT_THREADS) {          // all of a thread's loads in flight at once
        unsigned v[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int e = e0 + q * PT_THREADS;
            v[q] = e < total ? __hip_atomic_load(&A.swap_part[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int e = e0 + q * PT_THREADS;
            if (v[q]) atomicAdd(&s_n[e % NP], v[q]);
        }
    }
    __syncthreads();
    for (int e = tid; e < A.nranks * NP; e += PT_THREADS) {
        const int q = e / NP, j = e - q * NP;                        // ext pair j+1 = global pair (rung_begin+j+1, rung_begin+j)
        const PipeBox bx = pipe_box(A.boxes[q], T, W, D);
        __hip_atomic_store(bx.counts + (size_t)(A.sweep & 3u) * T + (A.rung_begin + j), s_n[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid < A.nranks) pipe_raise(pipe_box(A.boxes[tid], T, W, D).flags + PF_CNT0 + A.rank, A.sweep + 1);
    if (A.stats && tid == 0) {            // debug: collector wait / tail, wall-clock ticks
        atomicAdd(A.stats + 10, (unsigned long long)(dbg_t1 - dbg_t0));
        atomicAdd(A.stats + 11, 1ull);
        atomicAdd(A.stats + 12, (unsigned long long)(wall_clock64() - dbg_t1));
        atomicAdd(A.stats + 13, 1ull);
    }
}

inline __global__ __launch_bounds__(256) void k_pipe_bottom(const PipeArgs A) {
    __shared__ int32_t s_src[PIPE_COLS];
    __shared__ int32_t s_below[PIPE_COLS];
    pipe_bottom_block(A, blockIdx.x * PIPE_COLS, s_src, s_below, nullptr, nullptr);
}

// ---- hens_pipe_selftest: the three access patterns the pipeline relies on, between two processes ----------
inline __global__ void k_probe_put(double* peer_buf, unsigned* peer_flag, int n, double tag) {
    for (int i = threadIdx.x; i < n; i += blockDim.x) sys_store(peer_buf + i, tag + i);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) pipe_raise(peer_flag, 1u);
}
inline __global__ void k_probe_fill(double* buf, int n, double tag) {     // system-scope stores into my own cached memory
    for (int i = threadIdx.x; i < n; i += blockDim.x) sys_store(buf + i, tag + i);
}
inline __global__ void k_probe_check(const unsigned* my_flag, const double* my_buf, const double* peer_cached, int n,
                              double tag_in, double tag_pull, int check_put, int check_pull, long long budget,
                              unsigned* result) {
    __shared__ unsigned bad;
    if (threadIdx.x == 0) {
        bad = 0;
        if (check_put) pipe_spin(my_flag, 1u, budget, result);     // result bit FLAG_PIPE_TIMEOUT on timeout
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        if (check_put && sys_load(my_buf + i) != tag_in + i) atomicOr(&bad, 1u);
        if (check_pull && sys_load(peer_cached + i) != tag_pull + i) atomicOr(&bad, 2u);
    }
    __syncthreads();
    if (threadIdx.x == 0 && bad) atomicOr(result, bad << 8);
}

}  // namespace hens
