/*
 * hipensemble.h - C ABI of libhipensemble.so, an MI355X (gfx950) native stepping
 * engine for Eryn's walker-parallel stretch move + parallel-tempering path.
 *
 * The reference (mikekatz04/Eryn v1.2.6) is pure Python and has no FFI for this
 * path; its plugin boundary is the Python `Move.propose(model, state)` protocol.
 * Each entry point below therefore cites the reference *Python* interface it
 * replaces (paths relative to /root/reference/src/eryn).  The Python side of the
 * boundary (eryn_amd/moves/stretch.py, eryn_amd/moves/tempering.py) binds these
 * with ctypes; INTEGRATION.md shows the stub an Eryn maintainer would add.
 *
 * Conventions
 *   - every function returns an hens_status (0 = ok, < 0 = error);
 *     hens_last_error(ctx) gives the message (ctx may be NULL after a failed
 *     hens_create).
 *   - host buffers are caller-owned, C-contiguous, little-endian; the library
 *     copies and never retains host pointers.
 *   - layouts are the reference's, with nleaves_max == 1 squeezed away:
 *       x[ntemps][nwalkers][ndim] f64, logl/logp[ntemps][nwalkers] f64,
 *       betas[ntemps] f64.  A context that owns the ladder shard
 *       [rung_begin, rung_end) exchanges only those rungs (Tl = rung_end -
 *       rung_begin rows) through x/logl/logp; betas is always the full ladder.
 *   - one host thread per context; no callbacks into the caller.
 */
#ifndef HIPENSEMBLE_H
#define HIPENSEMBLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct hens_ctx hens_ctx;

typedef enum hens_status {
    HENS_OK = 0,
    HENS_ERR_INVALID = -1,          /* bad argument / shape            -> ValueError   */
    HENS_ERR_HIP = -2,              /* HIP runtime failure             -> RuntimeError */
    HENS_ERR_STATE = -3,            /* call order / missing setup      -> RuntimeError */
    HENS_ERR_TOO_FEW_WALKERS = -4,  /* red_blue.py:108-114             -> RuntimeError */
    HENS_ERR_NONFINITE = -5,        /* ensemble.py:1258-1262,1541-1542 -> ValueError   */
    HENS_ERR_UNSUPPORTED = -6       /* feature outside the hot path    -> NotImplementedError */
} hens_status;

typedef enum hens_likelihood {
    HENS_LIKE_GAUSS_DENSE = 0,      /* -0.5 (x-mu)^T P (x-mu), tests/test_eryn.py:33-35 */
    HENS_LIKE_GAUSS_DIAG = 1,       /* same with diagonal precision (test_base's identity covariance) */
    HENS_LIKE_ROSENBROCK = 2,       /* -(sum b (x[i+1]-x[i]^2)^2 + (a-x[i])^2), BASELINE config 5 */
    HENS_LIKE_HOST = 3,             /* arbitrary caller-side log_like_fn: hens_propose_split / hens_accept_split */
    HENS_LIKE_TEMPLATE = 4          /* variable-dimension template model on leaf-packing records: hens_rj_*  */
} hens_likelihood;

/* Construction parameters.  Mirrors the keyword arguments that reach the path:
 * EnsembleSampler(nwalkers, ndims, ...)           ensemble.py:211-247
 * StretchMove(a=2.0, live_dangerously=False)      moves/stretch.py:37, moves/red_blue.py:41-47
 * TemperatureControl(adaptive, adaptation_lag,    moves/tempering.py:242-255
 *                    adaptation_time, stop_adaptation)
 * fill_zero_leaves_val = -1e300                   ensemble.py:242,1486-1513          */
typedef struct hens_config {
    int32_t ntemps;            /* full ladder length T                                   */
    int32_t nwalkers;          /* W                                                      */
    int32_t ndim;              /* D                                                      */
    int32_t rung_begin;        /* ladder shard owned by this context: [rung_begin,       */
    int32_t rung_end;          /*   rung_end); 0, ntemps for a single GPU                */
    int32_t device_id;         /* HIP device ordinal                                     */
    int32_t likelihood_kind;   /* hens_likelihood                                        */
    int32_t tempered;          /* 0: logP = logl + logp (move.py:443-457); 1: betas      */
    int32_t live_dangerously;  /* skip the W >= 2 D guard (red_blue.py:108-114)          */
    int32_t adaptive;          /* ladder adaptation on (tempering.py:632-633)            */
    int32_t adaptation_delay;  /* ladder pipeline only: 0 = the reference's schedule (the swap ratios of sweep s move the
                                * ladder before iteration s+1 - every rank then waits for the whole cascade);
                                * 1 = they move it before iteration s+2, which lets the ranks pipeline          */
    int32_t ndim_active;       /* 0: = ndim.  Otherwise the number of REAL parameters in rows that the caller padded to a
                                * width the compile-time-width kernels serve (8, 16, 32, 64, 128): the pads hold zeros,
                                * lie in a (-inf, +inf) prior interval and meet zero rows / columns of the precision
                                * matrix, so they never change; only the Hastings factor (ndim - 1) log zz
                                * (stretch.py:223) and the W >= 2 ndim guard (red_blue.py:108-114) count real
                                * parameters.  eryn_amd.engine.HipEnsemble pads on upload and strips on download. */
    int64_t stop_adaptation;   /* < 0: never stop (tempering.py:591)                     */
    double a;                  /* stretch scale                                          */
    double fill_value;         /* likelihood of walkers outside the prior support        */
    double adaptation_lag;     /* 10000                                                  */
    double adaptation_time;    /* 100                                                    */
    uint64_t seed;             /* Philox key for hens_step                               */
} hens_config;

/* Lifetime.  Replaces: State / Branch buffer allocation (state.py:330-562). */
int hens_create(const hens_config* cfg, hens_ctx** out);
void hens_destroy(hens_ctx* ctx);
const char* hens_last_error(const hens_ctx* ctx);
int hens_synchronize(hens_ctx* ctx);

/* Model constants.
 * hens_set_prior_box: ProbDistContainer({i: uniform_dist(lo_i, hi_i)}) (prior.py:12-91,
 *   337-392).  logp_inside is sum_d log(1/(hi_d-lo_d)) accumulated by the caller in the
 *   reference's order, so an in-support prior value is bit-identical.
 * hens_set_gaussian: the (mu, invcov) `args` of the reference tests' log_like_fn
 *   (tests/test_eryn.py:33-35,100-104).  prec has D*D entries (dense) or D (diag).
 * hens_set_rosenbrock: constants of the config-5 stress likelihood. */
int hens_set_prior_box(hens_ctx* ctx, const double* lo, const double* hi, double logp_inside);
/* hens_set_periodic: the `periodic` argument of EnsembleSampler / Move (ensemble.py:165-168,338-347,528-536; a
 *   PeriodicContainer, utils/periodic.py:11-151) for the single branch: period[d] > 0 makes parameter d periodic - the
 *   stretch move measures c - s the short way round (stretch.py:136-141 -> periodic.py:49-116) and every proposal is
 *   wrapped into [0, period) with NumPy's remainder (stretch.py:149-154, gaussian.py:110-115 -> periodic.py:118-151);
 *   0 = not periodic; NULL or all zeros = no periodic parameters.  Every entry point honours it (the one- and two-launch
 *   iterations' compile-time-width kernels and the generic-width kernel alike, and the ranks of a ladder pipeline - set it
 *   on every rank BEFORE hens_pipe_init); not available on a leaf-packing context. */
int hens_set_periodic(hens_ctx* ctx, const double* period);
int hens_set_gaussian(hens_ctx* ctx, const double* mu, const double* prec);
int hens_set_rosenbrock(hens_ctx* ctx, double a, double b);

/* State transfer.  Replaces State(coords, log_like=, log_prior=, betas=) construction and
 * the State snapshot handed to Backend.save_step (state.py:437-517, backends/backend.py:1014-1091).
 * Any output pointer may be NULL.  logl/logp NULL on upload = "not evaluated yet". */
int hens_upload_state(hens_ctx* ctx, const double* x, const double* logl, const double* logp,
                      const double* betas);
int hens_download_state(hens_ctx* ctx, double* x, double* logl, double* logp, double* betas);

/* Initial log-prior / log-likelihood of the resident coordinates.
 * Replaces EnsembleSampler.compute_log_prior + compute_log_like as called from
 * sample() (ensemble.py:898-912; 1127-1217; 1219-1545 incl. the -inf-prior skip and
 * the fill value).  Returns HENS_ERR_NONFINITE if a coordinate is inf/NaN. */
int hens_eval_state(hens_ctx* ctx);

/* One red/blue half-step on all resident rungs, driven by caller-supplied draws
 * ("parity mode").  Replaces one trip of the `for split in range(nsplits)` body of
 * RedBlueMove.propose (red_blue.py:148-323): StretchMove.get_proposal (stretch.py:160-231),
 * compute_log_prior, compute_log_like, the MH test (red_blue.py:292-294) and Move.update
 * (move.py:472-703).
 *   labels[Tl][W]   u8   split label of every walker after the reference's shuffle (red_blue.py:121-124)
 *   rint[Tl][Ns]    i64  R.randint(Nc, size=(T, Ns))                  (stretch.py:93-99)
 *   u_zz[Tl][Ns]    f64  R.rand(T, Ns) for the stretch factor         (stretch.py:129-132)
 *   u_acc[Tl][Ns]   f64  R.rand(T, Ns) for the accept test            (red_blue.py:294)
 *   keep_out[Tl][Ns] u8  accept mask in the order of the ascending moving set S
 * Calls must alternate split = 0, 1.  Synchronous. */
int hens_stretch_split(hens_ctx* ctx, int32_t split, const uint8_t* labels, const int64_t* rint,
                       const double* u_zz, const double* u_acc, uint8_t* keep_out);

/* Number of sets of the red-blue move the parity API runs (`RedBlueMove(nsplits=...)`, red_blue.py:41-47,148: walker w of a rung
 * starts in set w % nsplits, shuffled per rung; set k moves against the other sets concatenated in set order, stretch.py:199).
 * Default 2.  With nsplits = n the split calls of one move run 0 .. n-1 in order, labels take values in [0, n), set k holds
 * ceil((W - k) / n) walkers and rint indexes the W - that many others.  hens_step draws the n sets itself (labels = a keyed
 * permutation mod n; round 4), except on a ladder shard, which keeps two sets. */
/* StretchMove.a as a mutable attribute (stretch.py:37; the reference's tuning hook mutates move.a, utils/updates.py:130-175):
 * the scale of every proposal made after the call, in both RNG modes. */
int hens_set_stretch_scale(hens_ctx* ctx, double a);
int hens_set_nsplits(hens_ctx* ctx, int32_t nsplits);

/* Host-callable likelihood (contexts created with HENS_LIKE_HOST).  The half-step of
 * hens_stretch_split is cut in two around the caller's log_like_fn (ensemble.py:1219-1545,
 * 1623-1667): hens_propose_split returns the proposed points q[Tl][Ns][D] and inbox[Tl][Ns]
 * (1 = inside the prior box; walkers with 0 must not be evaluated and get the fill value,
 * ensemble.py:1279-1282,1486-1513); hens_accept_split takes logl[Tl][Ns] and the accept uniforms and
 * performs the MH test and Move.update on the device.  Same alternation rule as hens_stretch_split;
 * hens_eval_state on such a context fills log_prior only.  hens_step is not available. */
int hens_propose_split(hens_ctx* ctx, int32_t split, const uint8_t* labels, const int64_t* rint,
                       const double* u_zz, double* q_out, uint8_t* inbox_out);
int hens_accept_split(hens_ctx* ctx, int32_t split, const double* logl, const double* u_acc,
                      uint8_t* keep_out);

/* Hot->cold swap cascade + ladder adaptation, driven by caller-supplied draws.
 * Replaces TemperatureControl.temper_comps (tempering.py:598-649): temperature_swaps
 * (:484-561), do_swaps_indexing (:351-482), adapt_temps (:585-596).
 *   iperm, i1perm [T-1][W] i64; u_swap [T-1][W] f64: row j holds the draws of the pair
 *   (i, i-1), i = T-1-j, i.e. the order np.random.permutation / uniform are called.
 *   adapt: 0 = swaps only (rj.py:381-382 calls temper_comps(adapt=False)).
 *   sel_out [T-1][W] u8 (same row order), swaps_accepted_out [T-1] indexed by i-1.
 * Requires the whole ladder resident (rung_begin = 0, rung_end = ntemps).  Synchronous. */
int hens_pt_sweep(hens_ctx* ctx, const int64_t* iperm, const int64_t* i1perm, const double* u_swap,
                  int32_t adapt, uint8_t* sel_out, double* swaps_accepted_out);

/* Production stepping: n_iters iterations of (split 0, split 1, PT cascade, adaptation)
 * with device-side Philox4x32-10 draws, no host round trip.  Replaces the body of the
 * hot loop of EnsembleSampler.sample for one in-model StretchMove (ensemble.py:965-981).
 * Asynchronous on the context's stream; hens_synchronize / any download waits. */
int hens_step(hens_ctx* ctx, int64_t n_iters);

/* hens_step(n_before), keep the accept counters as they stand on the device, hens_step(n_last): the reference stores the
 * accept mask of the LAST thinned sub-iteration only (ensemble.py:968-979: `accepted` is re-zeroed per sub-iteration), and
 * a host loop with thin_by > 1 needs no call split and no counter read in between.  hens_get_marked_counters downloads the
 * kept counts ([rungs][nwalkers] f64 each: stretch move, MH move - zeros without one; either pointer may be null). */
int hens_step_marked(hens_ctx* ctx, int64_t n_before, int64_t n_last);
int hens_get_marked_counters(hens_ctx* ctx, double* accepted, double* accepted_mh);
/* hens_step(n_iters) + what EnsembleSampler's loop reads after every proposal (ensemble.py:974-977: `accepted_out`,
 * `move.temperature_control.swaps_accepted`; tempering.py:563-649: the adapted ladder), in one call and one small copy:
 *   accepted_last[Tl][W] u8   accept counts of the call's last n_last iterations (one sampler sub-iteration =
 *                             num_repeats_in_model proposals; saturates at 255)
 *   swaps_last[T-1], betas[T] of the last cascade / after its adaptation.   Any pointer may be NULL.
 * The walkers are NOT copied: the drop-in moves return a State whose arrays download on first read
 * (eryn_amd/state.py: DeviceState; SURVEY 8 b-2 "device-resident State mirror"). */
int hens_step_report(hens_ctx* ctx, int64_t n_iters, int64_t n_last, uint8_t* accepted_last, double* swaps_last, double* betas);

/* Counters.  Replaces Move.accepted / num_proposals (move.py:404-421, red_blue.py:326-327),
 * TemperatureControl.swaps_accepted / time (tempering.py:542,596).  Any pointer may be NULL.
 *   accepted[Tl][W] f64 cumulative, swaps_last[T-1], swaps_total[T-1] f64. */
int hens_get_counters(hens_ctx* ctx, double* accepted, int64_t* num_proposals,
                      double* swaps_last, double* swaps_total, int64_t* adapt_time);
int hens_reset_counters(hens_ctx* ctx);
int hens_set_adapt_time(hens_ctx* ctx, int64_t t);

/* Timing of the most recent hens_step call.  hens_set_profiling(ctx, mode): 0 off; 1 a HIP event pair around every launch
 * (forces the call onto the HIP stream); 2 (round 6) the launches' own dispatch timestamps on the queue the call uses anyway -
 * on one GPU the context's AQL queue, same packets, fences and kernel arguments as an untimed call, each packet with a
 * completion signal the packet processor stamps (hsa_amd_profiling_get_dispatch_time: what rocprofv3's kernel trace reads);
 * calls that step on the HIP stream whatever the mode (pipeline ranks, the Gaussian move of a mix) fall back to event pairs.
 * hens_timing::clock says which one the figures came from.
 *   total_ms      wall time of the whole call on the device (0 unless per-kernel profiling is on or HENS_STEP_EVENTS=1: the
 *                 event pair costs a short call two barrier packets)
 *   stretch_ms    summed duration of the stretch kernels, n_stretch = their count
 *   pt_ms         summed duration of the PT cascade kernels, n_pt = their count
 * Per-kernel figures are only filled when per-kernel events were enabled with
 * hens_set_profiling(ctx, 1) (they serialise the stream slightly). */
typedef struct hens_timing {
    double total_ms;
    double stretch_ms;
    double pt_ms;
    double plan_ms;
    int64_t n_stretch;
    int64_t n_pt;
    int64_t n_plan;
    int64_t n_iters;
    double fused_ms;          /* launches that run the second half-step and the cascade together (k_split1_pt) */
    int64_t n_fused;
    int64_t clock;            /* 0 none, 1 HIP event pairs on the HIP stream, 2 dispatch timestamps of the AQL packets */
} hens_timing;
int hens_set_profiling(hens_ctx* ctx, int32_t per_kernel_events);
int hens_get_timing(hens_ctx* ctx, hens_timing* out);
/* Dev aid: begin / end (us after the first launch's begin, from the dispatch packets' own timestamps) of every launch of the
 * last hens_step call that ran with per-kernel events; *n_out = values available (2 per launch). */
int hens_debug_launch_times(hens_ctx* ctx, double* out_us, int64_t capacity, int64_t* n_out);

/* Ladder sharding (one context per GPU, rungs [rung_begin, rung_end)).  The stretch
 * step needs no communication (complement walkers are drawn within a rung,
 * red_blue.py:183-193).  The PT cascade (tempering.py:484-561) is replayed by EVERY rank from the
 * all-gathered log-likelihoods: in column form it is one cheap parallel kernel, and identical
 * decisions / betas on all ranks need no further agreement.  Only walker rows that change rank
 * travel (an all-to-all of [dest id | x | logp] records).  Protocol per iteration, see
 * eryn_amd/ladder.py:
 *   hens_stretch_iter (or two hens_stretch_split calls)     local, no communication
 *   all-gather  device_buffers.logl -> device_buffers.gather_logl           (RCCL)
 *   hens_pt_plan_sharded     decisions, betas, send buffer packed, per-peer counts returned
 *   all-to-all  send_rows -> recv_rows with those counts                    (RCCL)
 *   hens_pt_finish_sharded   received rows scattered into free pool slots, buffers swapped */
typedef struct hens_device_buffers {
    void* logl;          /* f64 [Tl][W] resident rungs (current buffer; changes every PT step) */
    void* gather_logl;   /* f64 [T][W] staging for the all-gathered ladder                     */
    void* send_rows;     /* f64 [row_capacity][D + 2], segments in peer order                  */
    void* recv_rows;     /* f64 [row_capacity][D + 2]                                          */
    int64_t row_capacity;
    int64_t row_doubles; /* D + 2                                                              */
    void* stream;        /* hipStream_t the library launches on                                */
} hens_device_buffers;
int hens_get_device_buffers(hens_ctx* ctx, hens_device_buffers* out);
int hens_set_stream(hens_ctx* ctx, void* hip_stream);

/* One Philox iteration of both red/blue halves on the resident rungs, no PT, asynchronous.
 * Same kernels and draws as hens_step; exists so a sharded ladder can interleave communication. */
int hens_stretch_iter(hens_ctx* ctx);

/* Sharded PT, step 1.  iperm/i1perm/u_swap as in hens_pt_sweep, or all NULL for device-side
 * Philox draws (identical on every rank: they depend only on seed and iteration).
 * rank_of_rung[T] gives the owner of every rung; this context is rank `my_rank`.
 * Outputs (host): send_counts[nranks], recv_counts[nranks] in rows; sel_out / swaps_accepted_out
 * as in hens_pt_sweep (may be NULL).  On return the send buffer is packed.  Synchronous. */
int hens_pt_plan_sharded(hens_ctx* ctx, const int64_t* iperm, const int64_t* i1perm,
                         const double* u_swap, int32_t adapt, const int32_t* rank_of_rung,
                         int32_t nranks, int32_t my_rank, int64_t* send_counts, int64_t* recv_counts,
                         uint8_t* sel_out, double* swaps_accepted_out);
/* Sharded PT, step 2: scatter n_recv received rows (recv_rows, any order) and swap buffers. */
int hens_pt_finish_sharded(hens_ctx* ctx, int64_t n_recv);

/* ---- Metropolis-Hastings proposals: GaussianMove / MHMove ---------------------------
 * One full-ensemble proposal q = x + step for every walker of every resident rung, box prior,
 * likelihood, tempered accept test with factors = 0, update - MHMove.propose (mh.py:56-193) with
 * GaussianMove.get_proposal (gaussian.py:68-195).  The PT sweep that ends the reference's propose()
 * (mh.py:190-191) is hens_pt_sweep, as for the stretch move.
 *   step  f64 [Tl][W][D]  q - x: factor * scale * randn (gaussian.py:166-167) or the multivariate-normal
 *                         draw (gaussian.py:265-268), zero where mode="random"/"sequential" keeps a coordinate
 *   u_acc f64 [Tl][W]     accept uniforms (mh.py:157)
 *   keep_out u8 [Tl][W]   accept mask, or NULL */
int hens_mh_step(hens_ctx* ctx, const double* step, const double* u_acc, uint8_t* keep_out);
/* Device-side draws for hens_step: with probability `weight` an iteration is a Gaussian MH proposal
 * instead of the stretch move (the reference's weighted move mix, ensemble.py:971).  kind 0: isotropic,
 * scale[1] = standard deviation; 1: axis-aligned, scale[D] = standard deviations; 2: full covariance,
 * scale[D*D] = lower Cholesky factor, row-major.  kind < 0 switches the mix off. */
int hens_set_mh_proposal(hens_ctx* ctx, int32_t kind, const double* scale, double weight);
/* accept counts [Tl][W] and number of MH proposals so far (Move.accepted / num_proposals of the MH move). */
int hens_get_mh_counters(hens_ctx* ctx, double* accepted, int64_t* num_proposals);

/* ---- Ladder pipeline: sharded stepping by neighbour exchange (one process per GPU) ------------------
 * No reference counterpart (the reference has no distributed path; it walks the whole ladder in one
 * process, tempering.py:598-649).  Each rank keeps a contiguous rung range (rank 0 = coldest rungs)
 * and a MAILBOX in uncached device memory that its neighbours write into directly (one-sided stores
 * over xGMI through HIP IPC) followed by a flag the consuming kernel spins on; rows that move up a
 * boundary are read straight out of the cold neighbour's walker pool (also IPC-mapped).  After
 *   hens_pipe_init           allocate the mailbox, export the IPC handles (mailbox + pool)
 *   (exchange the blobs: torch.distributed all_gather in eryn_amd.ladder)
 *   hens_pipe_connect        map every rank's mailbox
 * hens_step(n) on every rank advances the WHOLE ladder by n iterations with device-side draws:
 * the result is bit-identical to one context holding all the rungs.  A neighbour that stops
 * answering makes the waiting rank fail with HENS_ERR_STATE after HENS_PIPE_TIMEOUT_S (default 20 s)
 * instead of hanging the GPU. */
#define HENS_PIPE_BLOB_BYTES 128   /* two HIP IPC handles: the rank's mailbox and its walker pool */
int hens_pipe_init(hens_ctx* ctx, int32_t nranks, int32_t my_rank, void* blob_out /* HENS_PIPE_BLOB_BYTES or NULL */,
                   int64_t* mailbox_bytes_out);
int hens_pipe_connect(hens_ctx* ctx, const void* blobs /* nranks * HENS_PIPE_BLOB_BYTES, rank order */);
/* ---- Staged transport: the pipeline's messages over RCCL send/recv ------------------------------------
 * Same kernels, same protocol, but the stores go into LOCAL outboxes and the caller moves the message regions
 * between the stages with point-to-point sends (torch.distributed isend/irecv = grouped ncclSend/ncclRecv on
 * ROCm; eryn_amd.ladder.StagedPipeline).  For nodes where peer mappings are unavailable, and as the literal
 * "RCCL neighbour exchange" of the design brief; it costs host-ordered launches and dense (not sparse) row
 * messages, so the one-sided transport above stays the default.  Per iteration:
 *   hens_pipe_regions                      (pointers of this sweep's message regions)
 *   hens_pipe_stage 0                      move
 *   send ldn_out + ldn_rows_out up,  recv ldn_in + ldn_rows_in from below
 *   recv lup_in from above
 *   hens_pipe_stage 1                      walk
 *   send lup_out down
 *   recv rows_in from above                (before the bottom kernel: a walker may fall through all my rungs)
 *   hens_pipe_stage 2                      bottom
 *   send rows_out down
 *   all-reduce(sum) cnt_out over the ranks, copy to cnt_in */
typedef struct hens_pipe_region_table {
    void* ldn_out;  void* ldn_rows_out;  void* ldn_in;  void* ldn_rows_in;   /* f64: lp_doubles, row_doubles each */
    void* lup_out;  void* lup_in;                                           /* f64: lp_doubles                  */
    void* rows_out; void* rows_in;                                          /* f64: row_doubles                 */
    void* cnt_out;  void* cnt_in;                                           /* u32: cnt_words                   */
    int64_t lp_doubles, row_doubles, cnt_words;
    void* stream;
} hens_pipe_region_table;
int hens_pipe_connect_staged(hens_ctx* ctx);
/* ---- The same with the messages sent by the LIBRARY over RCCL (hens_comm_init / hens_comm_destroy) -------------
 * hens_comm_unique_id: ncclGetUniqueId on ONE rank (128 bytes out); the caller hands the id to the other ranks.
 * hens_comm_init: collective; every rank of the ladder calls it with the same id.  The context becomes a rank of the staged
 *   pipeline (hens_pipe_init + hens_pipe_connect_staged if not done yet) whose neighbour exchanges - grouped ncclSend / ncclRecv
 *   with the ladder neighbours, one all-reduce of the swap counts per sweep - the library enqueues itself on the context's stream:
 *   hens_step(ctx, n) is then ONE call for n iterations (the reference has no distributed path: tempering.py:484-561 is the
 *   cascade being sharded).  librccl.so.1 is dlopen()ed (in a PyTorch-ROCm process: torch's own copy).
 * hens_comm_selfsend: dev aid - n doubles through ncclSend / ncclRecv to the calling rank itself (a one-GPU test of the transport). */
int hens_comm_unique_id(void* unique_id_out /* 128 bytes */);
int hens_comm_init(hens_ctx* ctx, int32_t nranks, int32_t rank, const void* unique_id /* 128 bytes */);
int hens_comm_destroy(hens_ctx* ctx);
int hens_comm_selfsend(hens_ctx* ctx, int64_t n, const double* src_host, double* dst_host);
int hens_pipe_regions(hens_ctx* ctx, hens_pipe_region_table* out);
int hens_pipe_stage(hens_ctx* ctx, int32_t stage);

/* Self-test of the peer accesses the pipeline relies on (one-sided put + flag into uncached memory, pull out of
 * ordinary device memory), between this process and its ladder neighbours' processes, which find each other
 * through files in `dir`.  Needs no context.  eryn_amd.ladder runs it in a throw-away process per rank before
 * hens_pipe_connect, so a node where peer mappings do not work fails here and not in the sampler. */
int hens_pipe_selftest(int32_t device_id, int32_t rank, int32_t nranks, const char* dir, double timeout_s);
/* Debug (env HENS_PIPE_STATS=1 at hens_pipe_init): where the pipeline waits.  out16 = 8 pairs (wall-clock
 * ticks spent spinning, number of waits): [0] stretch prologue on arrived rows, [1] stretch prologue on swap
 * counts, [2] walk on the hot neighbour's columns, [3] bottom on the cold neighbour's rung, [4] bottom on rows
 * from above.  Ticks are hipDeviceAttributeWallClockRate (100 MHz on MI355X). */
int hens_pipe_debug_stats(hens_ctx* ctx, uint64_t* out16, int32_t reset);
/* Same, for contexts that live in ONE process (tests; several shards on one GPU): peers[nranks]. */
int hens_pipe_connect_local(hens_ctx* ctx, hens_ctx* const* peers);

/* Debug: per-workgroup phase timestamps of the stretch kernel (shader-clock ticks, 8 per
 * workgroup: start, A done, barrier, B done, barrier, C done, D done, end).  enable != 0 makes
 * the following stretch launches record; enable == 0 copies the last launch's trace to `out`
 * (capacity in 64-bit words) and stops recording.  *n_out = words written. */
int hens_debug_trace(hens_ctx* ctx, int32_t enable, uint64_t* out, int64_t capacity, int64_t* n_out);

/* Debug: the keyed pseudo-random permutation the Philox mode uses for iteration `iter`:
 * which = 0 the PT column map of global rung `rung`, which = 1 the split labelling permutation of
 * rung `rung`.  out[nwalkers] i32. */
int hens_debug_permutation(hens_ctx* ctx, int32_t which, int32_t rung, int64_t iter, int32_t* out);

/* Reversible-jump leaf packing (BASELINE config 4).  A context created with HENS_LIKE_TEMPLATE holds
 * variable-dimension walkers as RECORDS of ndim doubles:
 *     [ branch 0: nleaves_max_0 x 3 coordinates | branch 1: ... | leaf mask of branch 0 | mask of branch 1 | ... | pad ]
 * where a mask is the reference's inds[t, w, :] (state.py:330-562, backends/backend.py:1049-1059) as an integer stored
 * in a double (bit n = leaf slot n in use).  hens_upload_state / hens_download_state / hens_eval_state / hens_pt_sweep /
 * hens_get_counters work on records unchanged: a PT swap carries all leaves and masks of a walker (tempering.py:376-480).
 *
 * hens_rj_set_model: the model of the reference's own RJ tests (tests/test_eryn.py:38-92, 341-507): per branch a leaf
 *   kind (0 Gaussian pulse a exp(-(t-b)^2 / 2c^2), 1 sine a sin(2 pi b t + c)), leaf budget [nleaves_min, nleaves_max],
 *   a uniform box prior per leaf parameter (lo / hi [nbranches][3]; leaf_logp[b] = sum_d log(1/(hi-lo)) accumulated by
 *   the caller in the reference's order, prior.py:364-383) and the data (t, y)[ndata], sigma of
 *   logL = -1/2 sum(((template - y) / sigma)^2).
 * hens_rj_mh_step: the in-model GaussianMove on the packed active leaves of every branch (mh.py:56-193,
 *   gaussian.py:68-115,265-268) with the caller's draws: step[Tl][W][ncoord] in record layout (zero on unused slots),
 *   u_acc[Tl][W].  Replaces compute_log_prior / compute_log_like with inds (ensemble.py:1127-1217, 1219-1545,
 *   utils/utility.py:8-40 leaf grouping), the accept test and Move.update (move.py:472-703).
 * hens_rj_bd_step: DistributionGenerateRJ on one branch (distgenrj.py:35-222, rj.py:145-388): change[Tl][W] in
 *   {-1, 0, +1} after the edge rule (distgenrj.py:69-73), leaf[Tl][W] the slot that is born or dies, birth[Tl][W][3] the
 *   prior draw of a born leaf, u_acc[Tl][W].  The library adds the proposal factors -/+ log q(leaf), the edge factors
 *   (rj.py:236-270) and the fix_logp_gibbs rule (move.py:368-402).  Follow it with hens_pt_sweep(adapt = 0) (rj.py:381-382).
 * hens_rj_stretch_split: one half of the red / blue StretchMove on a state of several branches and leaves (round 5; SURVEY 8
 *   row a4's loop over branches): RedBlueMove.propose (red_blue.py:103-330) + StretchMove.get_proposal / choose_c_vals /
 *   get_new_points (stretch.py:74-231) without Gibbs sampling.  labels[Tl][W] in {0, 1} (arange(W) % 2 shuffled per rung,
 *   red_blue.py:119-124; the same array for both splits), rint[nbranches][Tl][Ns] - EVERY branch its own complement draw,
 *   an index into the other set in ascending walker order (stretch.py:93-100, 205) -, u_zz[Tl][Ns] the ONE stretch factor's
 *   uniform per walker (stretch.py:128-132), u_acc[Tl][Ns], keep_out[Tl][Ns] by position of the moving set (ascending walker
 *   order).  Every leaf slot of every branch moves, active or not; factors = (sum of nleaves_max * ndim - 1) log zz
 *   (stretch.py:222-223); the leaf masks stay and decide what prior and likelihood see; rows are updated in place (complements
 *   come from the other set).  Call split 0 then split 1, then hens_pt_sweep.  Fewer than twice as many walkers as leaf
 *   coordinates -> HENS_ERR_TOO_FEW_WALKERS (red_blue.py:103-114).
 * hens_rj_set_mh_scale + hens_rj_step: production: n iterations of (in-model move, swaps + adaptation, birth / death on
 *   a uniformly chosen branch, swaps) with device-side Philox draws of the same distributions.  Every walker's model at the
 *   data points stays resident and birth / death evaluates `model +- one leaf`; the resident models AND the log-likelihoods
 *   are re-evaluated from the coordinates whenever the state has crossed this interface since the last call (hens_upload_state,
 *   a parity-API move, hens_download_state) and whenever iteration % 64 == 63, so a log-likelihood carries the rounding of at
 *   most 63 iterations of +- updates (observed <= 6e-16 relative against the oracle, bar 1e-12) and a chain is a function of
 *   (State, seed, iteration counter, adaptation time): resumed from a downloaded State in a new context it is the uninterrupted
 *   chain bit for bit.
 * hens_rj_get_counters: accept counts of the birth / death move (hens_get_counters has the in-model move's). */
int hens_rj_set_model(hens_ctx* ctx, int32_t nbranches, const int32_t* kinds, const int32_t* nleaves_max,
                      const int32_t* nleaves_min, const double* lo, const double* hi, const double* leaf_logp,
                      int32_t ndata, const double* t, const double* y, double sigma);
/* A leaf-packing model WITHOUT a device likelihood (round 6): the reference's `ndims` per branch (ensemble.py:325-329) - 1 .. 4
 * box-prior parameters per leaf, up to 4 branches, up to 64 leaf slots and 128 record doubles in all - for chains whose likelihood is
 * the caller's function of the packed active leaves (ensemble.py:1306-1334, 1340-1545): such a context is stepped with
 * hens_rj_propose / hens_rj_accept only (hens_rj_step and the parity moves with the built-in template likelihood refuse it), and
 * hens_eval_state leaves the log-prior and the fill value as log-likelihood (the caller uploads its own).  Record layout as
 * hens_rj_set_model's with ndims[b] doubles per leaf slot; lo / hi: the branches' boxes one after the other; the `birth` arrays of
 * hens_rj_draws have a stride of max(ndims) doubles per walker. */
int hens_rj_set_model_general(hens_ctx* ctx, int32_t nbranches, const int32_t* ndims, const int32_t* nleaves_max,
                              const int32_t* nleaves_min, const double* lo, const double* hi, const double* leaf_logp);
int hens_rj_set_mh_scale(hens_ctx* ctx, const double* scale);
int hens_rj_mh_step(hens_ctx* ctx, const double* step, const double* u_acc, uint8_t* keep_out);
int hens_rj_bd_step(hens_ctx* ctx, int32_t branch, const int8_t* change, const int32_t* leaf, const double* birth,
                    const double* u_acc, uint8_t* keep_out);
int hens_rj_stretch_split(hens_ctx* ctx, int32_t split, const uint8_t* labels, const int64_t* rint, const double* u_zz,
                          const double* u_acc, uint8_t* keep_out);
int hens_rj_step(hens_ctx* ctx, int64_t n_iters);
int hens_rj_get_counters(hens_ctx* ctx, double* accepted_bd, int64_t* num_mh, int64_t* num_bd);

/* Debug / parity: everything hens_rj_step draws in iteration `iter` (a pure function of seed, iteration, global rung and
 * walker), in a form that maps onto the reference's draws so that the production leaf-packing path can be replayed through
 * the CPU oracle (tests/test_hip_rj.py):
 *   step [Tl][W][ind_off]   the in-model Gaussian step of every coordinate slot, record layout (gaussian.py:265-268; only the
 *                           active leaves' entries are consumed)
 *   u_mh, u_bd [Tl][W]      accept uniforms of the two moves (mh.py:157, rj.py:332)
 *   branch                  the branch of the birth / death move (ensemble.py:988-990, "separate_branches")
 *   coin [Tl][W] i8         +1 / -1 before the edge rule (distgenrj.py:63-73)
 *   sel  [Tl][W] u32        leaf selector: of `cnt` candidate slots in ascending order, the one of index (sel * cnt) >> 32
 *                           (distgenrj.py:97-112)
 *   birth [Tl][W][3]        coordinates of a leaf born in `branch` (generate_dist.rvs, prior.py:60-66)
 *   slot_* [T][W], uswap_* [T-1][W]   column maps and swap uniforms of the cascade after the in-model move and of the one
 *                           after the birth / death move, as in hens_debug_draws (tempering.py:526-541) */
int hens_rj_debug_draws(hens_ctx* ctx, int64_t iter, double* step, double* u_mh, int32_t* branch, int8_t* coin, uint32_t* sel,
                        double* birth, double* u_bd, int32_t* slot_mh, double* uswap_mh, int32_t* slot_bd, double* uswap_bd);

/* The between-model schedule of hens_rj_step (the sampler's rj_moves string, ensemble.py:434-480):
 *   0  "separate_branches" (default): one DistributionGenerateRJ per branch, one of them chosen per iteration
 *   1  "iterate_branches": ONE move that walks through every branch in turn (birth / death, accept, update per branch), then
 *      one sweep of swaps without adaptation; its accept counts are the last branch's (rj.py:169-388).
 *   2  "together" (ensemble.py:414-432): ONE proposal changes a leaf in EVERY branch of the walker - all coins and leaf choices,
 *      then the births branch by branch, the factors summed, one accept test (distgenrj.py:150-222).
 * With schedules 1 and 2 hens_rj_debug_draws returns branch = -1 and coin / sel / birth / u_bd for every branch in order
 * ([nbranches][Tl][W]...; the caller sizes them for nbranches either way; schedule 2: every row of u_bd is the one uniform). */
int hens_rj_set_schedule(hens_ctx* ctx, int32_t schedule);

/* Parity-mode birth / death over ALL branches in one proposal ("together"): change / leaf [nbranches][Tl][W],
 * birth [nbranches][Tl][W][3 - max(ndims) on a hens_rj_set_model_general context], ONE u_acc [Tl][W] (rj.py:145-388 with gibbs_sampling_setup = None). */
int hens_rj_bd_all_step(hens_ctx* ctx, const int8_t* change, const int32_t* leaf, const double* birth, const double* u_acc,
                        uint8_t* keep_out);

/* Leaf-packing moves with a HOST-CALLABLE likelihood (round 6).  Replaces, for states of several branches / leaves whose
 * log_like_fn is an arbitrary Python function, the proposal + prior half and the accept + update half of MHMove.propose
 * (mh.py:56-193), ReversibleJumpMove.propose (rj.py:145-388) and RedBlueMove.propose (red_blue.py:103-330); between the two
 * the caller does what EnsembleSampler.compute_log_like does on the host (ensemble.py:1306-1334, 1340-1545; utils/utility.py:
 * 8-40 groups_from_inds): packs the active leaves per branch, calls the user's function, fills -1e300 / fill_zero_leaves_val.
 * The leaf-packing twin of hens_propose_split / hens_accept_split.
 *   move   HENS_RJ_MOVE_MH       in-model Gaussian move on every active leaf: draws->step [Tl][W][ncoord], u_acc [Tl][W]
 *          HENS_RJ_MOVE_BD       birth / death on draws->branch: change, leaf [Tl][W], birth [Tl][W][3], u_acc [Tl][W]
 *          HENS_RJ_MOVE_BD_ALL   ... on every branch in one proposal: [nbranches][...] arrays as hens_rj_bd_all_step
 *          HENS_RJ_MOVE_STRETCH  one half of the red / blue stretch move: split, labels, rint, u_zz, u_acc as
 *                                hens_rj_stretch_split
 *   q_out[Tl][W][RW]   the proposed records (every slot's coordinates + the leaf masks, hens_upload_state's layout)
 *   logp_out[Tl][W]    their log-prior, Move.fix_logp_gibbs applied (move.py:368-402): -inf = do not evaluate
 *   moved_out[Tl][W]   1 for the walkers this proposal moves (all of them; a stretch half-step: the moving set)
 * hens_rj_accept(logl[Tl][W]) - the caller's log-likelihoods of the moved walkers (entries of the others are ignored; NaN ->
 * HENS_ERR_NONFINITE, ensemble.py:1542) - runs the tempered accept test and Move.update; keep_out[Tl][W] by WALKER. */
enum { HENS_RJ_MOVE_MH = 0, HENS_RJ_MOVE_BD = 1, HENS_RJ_MOVE_BD_ALL = 2, HENS_RJ_MOVE_STRETCH = 3 };
typedef struct hens_rj_draws {
    const double* step; const int8_t* change; const int32_t* leaf; const double* birth;
    const uint8_t* labels; const int64_t* rint; const double* u_zz; const double* u_acc;
    int32_t branch, split;
} hens_rj_draws;
int hens_rj_propose(hens_ctx* ctx, int32_t move, const hens_rj_draws* draws, double* q_out, double* logp_out, uint8_t* moved_out);
int hens_rj_accept(hens_ctx* ctx, const double* logl, uint8_t* keep_out);

/* The Philox iteration counter: the index of the NEXT iteration hens_step will run (iterations completed on this
 * context so far, by hens_step or by the parity API). */
int hens_get_iteration(hens_ctx* ctx, int64_t* iter_out);

/* Resume: set the Philox iteration counter.  The device draws are a pure function of (seed, iteration, global rung,
 * walker), so a chain continues bit-identically from an uploaded State when the counter (and the adaptation time,
 * hens_set_adapt_time) are restored to the values that State was taken at - the device-side form of the reference's
 * random_state checkpoint (backends/backend.py:1014-1091 stores R's state with every step, ensemble.py:605-647 restores
 * it).  Not on a connected pipeline rank (the ranks' sweep counters would have to move together). */
int hens_set_iteration(hens_ctx* ctx, int64_t iter);

/* Debug / parity: the Philox draws hens_step consumes in iteration `iter` (a pure function of seed, iteration,
 * global rung and walker), exported in a form that maps one-to-one onto the reference's draws so that a
 * production iteration can be replayed through the CPU oracle (tests/test_hip_replay.py):
 *   own, cw [Tl][W] i32   moving walker / its complement walker at every split position; positions < ceil(W/2)
 *                         belong to split 0 and list that half in ascending walker order like the reference's
 *                         boolean masks (red_blue.py:119-124,150-197), so labels[own] = (position >= ceil(W/2)) and
 *                         rint = index of cw in the ascending complement list (stretch.py:93-99)
 *   u_zz, u_acc [Tl][W]   the raw uniforms behind zz (stretch.py:129-132) and the accept test (red_blue.py:294)
 *   pt_slot [T][W] i32    slot of global rung t that cascade column c visits: for pair (i, i-1),
 *                         iperm[k=c] = pt_slot[i][c] and i1perm[k=c] = pt_slot[i-1][c] (tempering.py:526-541)
 *   u_swap [T-1][W]       row j = the uniforms of pair (T-1-j, T-2-j) in column order (tempering.py:535)
 *   is_mh                 1 if the weighted move choice of that iteration (ensemble.py:971) picks the MH move
 *   mh_step [Tl][W][D], mh_u [Tl][W]   the GaussianMove step rows (gaussian.py:166-167,265-268) and accept
 *                         uniforms (mh.py:157), if hens_set_mh_proposal was called
 * Any output pointer may be NULL.  Does not touch the sampler state. */
int hens_debug_draws(hens_ctx* ctx, int64_t iter, int32_t* own, int32_t* cw, double* u_zz, double* u_acc,
                     int32_t* pt_slot, double* u_swap, int32_t* is_mh, double* mh_step, double* mh_u);

/* Static description of the build. */
const char* hens_version(void);
int hens_device_count(void);
/* PCI address ("0000:c1:00.0") of HIP device `device_id` - the physical GPU behind an ordinal, whatever the visibility masks say.
 * Host-side helper of the ladder pipeline's one-rank-per-GPU check (eryn_amd/ladder.py; no reference counterpart: the reference is
 * single-process).  Returns HENS_OK and a NUL-terminated string in out[capacity], or a negative error code. */
int hens_device_pci_bus_id(int32_t device_id, char* out, int32_t capacity);

#ifdef __cplusplus
}
#endif
#endif /* HIPENSEMBLE_H */
