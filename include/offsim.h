/*
 * offsim.h -- C ABI of the MI355X-native Per-State Rejection Sampling (PSRS) engine.
 *
 * Drop-in boundary for the replay-loop hot path of microsoft/rl-offline-simulation (offsim4rl).
 * The reference has no FFI of its own (it is pure Python); each entry point below names the
 * reference function it replaces (paths relative to the reference checkout).  A maintainer binds
 * these with ctypes -- see INTEGRATION.md for the stub that goes into
 * offsim4rl/evaluators/per_state_rejection.py.
 *
 * Conventions
 *   - Every pointer is a DEVICE pointer (hipMalloc / torch.Tensor.data_ptr() of a contiguous
 *     ROCm tensor) unless the name ends in _host.  The library never frees caller memory and never
 *     allocates what it returns; scratch is passed in.
 *   - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream).  All
 *     calls are asynchronous with respect to the host and safe to capture in a hipGraph.
 *   - Return value: 0 = OFFSIM_OK, negative = error; offsim_last_error() gives the message for the
 *     calling thread.  Nothing throws across the boundary.  No Python state; call with the GIL released.
 *   - "slot" = z - z_base: latent states are stored as non-negative slots so that z = -1
 *     (CartpoleBoxEncoder failure code, offsim4rl/encoders/heuristic.py:23-24) is a legal queue key.
 */
#ifndef OFFSIM_H
#define OFFSIM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OFFSIM_OK 0
#define OFFSIM_EINVAL (-1)   /* bad argument */
#define OFFSIM_EHIP (-2)     /* HIP runtime error (launch, no device, ...) */
#define OFFSIM_EUNSUPPORTED (-3)

/* dtype tags */
#define OFFSIM_F32 0
#define OFFSIM_F64 1
#define OFFSIM_F16 2

/* accept/reject rule (offsim4rl/evaluators/per_state_rejection.py:97 `_reject` hook) */
#define OFFSIM_REJECT_DEFAULT 0 /* psrs.py:53-57  u > p_new[a]/p_log[a]/max(p_new/p_log)          */
#define OFFSIM_REJECT_NEVER 1   /* trivial_baselines.py:8-10,22-24  accept head, no RNG draw       */

/* arithmetic of the default rule, following NumPy promotion */
#define OFFSIM_PROB_F64 0 /* p_new f64 (p_log widened exactly): divisions and compare in f64      */
#define OFFSIM_PROB_F32 1 /* p_new f32 and p_log f32: divisions in f32, u rounded to f32          */

/* per-rollout status written by offsim_eval_mc / offsim_step_batch */
#define OFFSIM_ST_OK 0          /* step accepted / episode cap reached                              */
#define OFFSIM_ST_EXHAUSTED 1   /* PSRS.step returned (None,)*4: queue of current state empty       */
#define OFFSIM_ST_NO_INIT 2     /* PSRS.reset returned None: init queue empty                       */
#define OFFSIM_ST_KEYERROR 3    /* current state never occurs as a from-state (psrs.py:44)          */
#define OFFSIM_ST_INACTIVE 4    /* rollout had no current state (s is None); nothing done           */
#define OFFSIM_ST_PROTOCOL 5    /* internal: a bounded wait between the two wavefronts of offsim_eval_mc_streams expired
                                   (never expected; the rollout stops instead of hanging the stream) */

/* The logged-transition table, SoA, rows physically grouped by from-state (CSR).  Built by
 * offsim_group_by_state + offsim_table_gather from the OfflineDataset.experience arrays
 * (offsim4rl/data.py:46-58) and the encoder output (per_state_rejection.py:29-35).
 * Replaces PSRS._calculate_latent_state + the sorted/groupby of reset_sampler (psrs.py:16-17,26). */
typedef struct offsim_table {
    int64_t N;              /* logged transitions                                               */
    int32_t n_slots;        /* states are slots 0..n_slots-1                                    */
    int32_t nA;             /* actions                                                          */
    int32_t plog_dtype;     /* OFFSIM_F32 | OFFSIM_F64 | OFFSIM_F16                             */
    int32_t r_dtype;        /* OFFSIM_F32 | OFFSIM_F64                                          */
    const uint32_t *seg_off;  /* [n_slots+1] first grouped row of each state                    */
    const void *p_log;        /* [N,nA] logging-policy probabilities (hot candidate stream)     */
    const int32_t *a;         /* [N]    logged action            (hot candidate stream)         */
    const void *r;            /* [N]    reward                   (accept-only stream)           */
    const int32_t *z_next;    /* [N]    slot of the next state   (accept-only stream)           */
    const uint8_t *done;      /* [N]    terminal flag            (accept-only stream)           */
    const int32_t *orig_idx;  /* [N]    row in the caller's buffer (for accepted-index reports) */
    int64_t N0;               /* rows with step == 0 (all rows if `steps` is absent, data.py:72) */
    const int32_t *init_slot; /* [N0]   slot of the k-th initial row, buffer order (psrs.py:22) */
    const int32_t *init_orig; /* [N0]   its row in the caller's buffer                          */
    int64_t max_seg;          /* longest state segment (max of seg_off[s+1]-seg_off[s]); 0 = unknown:
                                 the shuffle then sizes its LDS for the worst case (65536 rows)      */
    int64_t min_seg;          /* shortest non-empty state segment; 0 = unknown (the shuffle then launches
                                 every size class)                                                  */
} offsim_table;

/* State of R independent simulated rollouts (one PSRS env each).  Owned by the caller. */
typedef struct offsim_rollouts {
    int32_t R;
    uint64_t *rng;          /* [R,4] rejection stream: PCG64 state hi, lo, inc hi, lo (psrs.py:20) */
    uint32_t *cursor;       /* [R,n_slots] candidates popped so far from each state's queue       */
    uint32_t *init_cursor;  /* [R] initial states popped so far (psrs.py:36)                      */
    int32_t *cur_slot;      /* [R] current state slot, -1 = none (self.s is None)                 */
    const uint32_t *perm;   /* queue order: perm[r*perm_stride + seg_off[s] + k] = grouped row of
                               the k-th element of state s's queue.  perm_stride = N for per-rollout
                               shuffles, 0 for one order shared by all rollouts; NULL = table order */
    int64_t perm_stride;
    const uint32_t *init_perm; /* init_perm[r*init_stride + k] = index into init_slot/init_orig   */
    int64_t init_stride;
    int32_t rng_kind;       /* OFFSIM_STREAM_*: what `rng` holds and which generator draws u (psrs.py:56)  */
} offsim_rollouts;
/* Provider of the rejection stream u ~ U (one draw per candidate examined, psrs.py:56):
 *   OFFSIM_STREAM_PCG64   NumPy's default_rng(seed).random(): u in [0,1).  The parity default: accepted-index sequences equal the
 *                         reference's bit for bit.  rng row = PCG64 state hi, lo, increment hi, lo (offsim_seed_streams).
 *   OFFSIM_STREAM_PHILOX  rocRAND's Philox4x32-10 through its device API (rocrand_init(seed, 0, 2 i) / rocrand): draw i of a rollout is
 *                         rocrand_uniform_double of the engine seeded with the rollout's seed, u in (0,1] -- a different, equally valid
 *                         sample path, NOT the reference's numbers; its oracle is the reference's own PSRS.step with
 *                         env.rejection_sampling_rng replaced by an object that replays this stream (tests/golden/make_golden.py).
 *                         rng row = seed, draws consumed so far, 0, 0.  Taken by offsim_step_batch, offsim_eval_mc, offsim_eval_td and
 *                         (round 6) by both compiled-policy scans, offsim_eval_mc_streams and offsim_eval_mc_keys, which fill their
 *                         draw rings from the same engine (rocrand_device::philox4x32_10_engine::ten_rounds, csrc/philox_dev.hpp).
 *                         In the compiled-policy
 *                         scans a draw is compared as the integer k = u * 2^53 in [1, 2^53] against the 53-bit key; k = 2^53 (u = 1.0
 *                         exactly, probability 2^-53 per draw) is looked at as 2^53 - 1, which differs from the reference's rule only
 *                         against an importance ratio of exactly 1 - 2^-53. */
#define OFFSIM_STREAM_PCG64 0
#define OFFSIM_STREAM_PHILOX 1

const char *offsim_last_error(void);
int offsim_version(void);
/* number of HIP devices visible, or a negative error; never initialises a context */
int offsim_device_count(void);

/* ---- table construction (a1, a6, a12) -------------------------------------------------------- */

/* Stable group-by of rows by slot: order[g] = original row of grouped row g, seg_off = CSR offsets.
 * == sorted(buffer, key=z) + groupby of psrs.py:26 (buffer order inside a state).
 * scratch: at least offsim_group_scratch_bytes(N, n_slots) bytes. */
int64_t offsim_group_scratch_bytes(int64_t N, int32_t n_slots);
int offsim_group_by_state(const int32_t *slot, int64_t N, int32_t n_slots, uint32_t *seg_off /*[n_slots+1]*/,
                          int32_t *order /*[N]*/, void *scratch, void *stream);

/* Gathers the caller's row-major arrays into the grouped SoA layout: dst[g] = src[order[g]].
 * elem_bytes in {1,2,4,8,...}; row_elems = elements per row (nA for p_log). */
int offsim_gather_rows(const void *src, const int32_t *order, int64_t N, int32_t row_bytes, void *dst, void *stream);

/* ---- sampler (a2, a3) ------------------------------------------------------------------------ */

/* np.random.default_rng(seed) for R seeds: SeedSequence -> PCG64 (psrs.py:20).  seeds, out on device. */
int offsim_seed_streams(const uint64_t *seeds, int32_t R, uint64_t *rng_out /*[R,4]*/, void *stream);

/* PSRS.reset_sampler(seed) queue shuffles (psrs.py:22-23,29-30) for n_perm seeds at once:
 * every state's queue and the init queue get a backward Fisher-Yates driven by a FRESH
 * default_rng(seed).  perm_out [n_perm,N] holds grouped rows, init_perm_out [n_perm,N0] indices. */
int offsim_shuffle_queues(const offsim_table *t, const uint64_t *seeds, int32_t n_perm, uint32_t *perm_out,
                          uint32_t *init_perm_out, void *stream);

/* PSRS.reset() (psrs.py:32-37) for every rollout with mask[r] != 0 (mask NULL = all): pops the
 * init queue, sets cur_slot; out_init_row[r] = caller-buffer row of the initial state or -1 (None). */
int offsim_env_reset(const offsim_table *t, offsim_rollouts *ro, const uint8_t *mask, int32_t *out_init_row,
                     void *stream);

/* ---- the replay loop (a4, a5, a7, a8, a9) ---------------------------------------------------- */

/* One PSRS.step(p_new) (psrs.py:39-51) per rollout with a current state.
 * p_new [R,nA] f64 (prob_mode F64) or f32 (prob_mode F32).
 * out_row[r]   caller-buffer row of the accepted transition, or -1
 * out_status[r] OFFSIM_ST_*;  out_popped[r] candidates consumed by this call.
 * max_pop > 0 caps the candidates popped (max_pop = 1 with OFFSIM_REJECT_NEVER is the "pop one
 * candidate for a Python-side _reject override" primitive); advance = 0 leaves cur_slot unchanged. */
int offsim_step_batch(const offsim_table *t, offsim_rollouts *ro, const void *p_new, int32_t prob_mode,
                      int32_t reject_mode, int32_t advance, int32_t *out_row, int32_t *out_status,
                      uint32_t *out_popped, void *stream);

/* ---- step server: PSRS.step for one environment without a launch per call (a4, a7) ----------------------------------
 * The reference's evaluator is driven by one Python call per simulated step (per_state_rejection.py:85-95); a kernel launch plus a
 * stream synchronise per call costs ~27 us against ~9 us for the reference's own Python step.  offsim_step_server_start leaves ONE
 * wavefront resident that serves offsim_step_batch's step (R = 1, same arithmetic, same state rows) for requests posted through a
 * mailbox in host-coherent pinned memory:
 *   host:   write p_new (p_head / p_tail) and cmd / reject_mode, then seq_in2 = previous seq_in + 1, THEN seq_in = the same
 *           wait until seq_out == seq_in, read row / status / popped
 *   device: ends on OFFSIM_SERVER_CMD_EXIT, or by itself after `idle_polls` polls (~1-2 us each) without a request: `state` says so,
 *           and the host starts it again.  While it runs, nothing else may touch the rollout's state rows (cursor, rng, cur_slot).
 * offsim_host_alloc / offsim_host_free: the mailbox's memory (hipHostMalloc, coherent + mapped: host and device see each other's
 * stores while the kernel runs).  `stream` must not be a stream the caller synchronises while the server is meant to stay up. */
#define OFFSIM_MAILBOX_MAX_ACTIONS 24
#define OFFSIM_SERVER_CMD_STEP 1     /* PSRS.step(p_new) */
#define OFFSIM_SERVER_CMD_POP_ONE 2  /* pop one candidate, accept it, leave the state (the Python-side _reject hook's primitive) */
#define OFFSIM_SERVER_CMD_EXIT 3
#define OFFSIM_SERVER_CMD_RESET 4    /* PSRS.reset (psrs.py:32-37): row = caller-buffer row of the initial state, or -1 */
#define OFFSIM_SERVER_STARTING 1
#define OFFSIM_SERVER_RUNNING 2
#define OFFSIM_SERVER_EXITED 3
typedef struct offsim_step_mailbox {
    /* the first 64 bytes are what ONE poll of the server reads */
    uint32_t seq_in;      /* host -> device: request number, written LAST */
    uint32_t cmd;         /* OFFSIM_SERVER_CMD_* */
    int32_t reject_mode;  /* OFFSIM_REJECT_* */
    uint32_t reserved0;
    double p_head[5];     /* p_new[0..4] (f64), or p_new[0..9] as packed f32 (OFFSIM_PROB_F32) */
    uint32_t reserved1;
    uint32_t seq_in2;     /* the request number once more, written BEFORE seq_in and behind everything else: a snapshot of the 64 bytes
                           * that shows the new number in both places holds the new payload, whatever order its parts were read in */
    double p_tail[OFFSIM_MAILBOX_MAX_ACTIONS - 5]; /* p_new[5..] (f64), or p_new[10..] as packed f32 */
    uint32_t reserved2[2];
    /* the answer: one 16-byte store of the device */
    uint32_t seq_out;     /* device -> host: the request served */
    int32_t row;          /* caller-buffer row of the accepted transition (RESET: of the initial state), or -1 */
    int32_t status;       /* OFFSIM_ST_* */
    uint32_t popped;      /* candidates consumed */
    uint32_t state;       /* OFFSIM_SERVER_* (0: never started) */
    uint32_t reserved3[3];
} offsim_step_mailbox;
int offsim_host_alloc(int64_t bytes, void **host_ptr);
int offsim_host_free(void *host_ptr);
int offsim_step_server_start(const offsim_table *t, offsim_rollouts *ro, offsim_step_mailbox *mailbox, int32_t prob_mode,
                             uint32_t idle_polls, void *stream);
/* The host side of ONE request, in C (no device call: stores to the mailbox, then a spin on seq_out): p_new = n_actions probabilities
 * (f64 / f32 by prob_mode; NULL for RESET), out3 = {row, status, popped}.  Returns OFFSIM_OK, or OFFSIM_SERVER_GONE (> 0) when the
 * server ended by itself before it saw the request -- the request stays posted: synchronise the server's stream, start it again (it
 * serves the posted request) and wait for seq_out == seq_in -- or OFFSIM_EHIP after max_spins polls (0: no bound) or, whatever
 * max_spins says, after OFFSIM_SERVER_ANSWER_SECONDS of wall-clock time spent with the server in state RUNNING (a dead server is
 * reported in seconds, not minutes; a launch still queued on a shared device is not timed; the environment variable
 * OFFSIM_SERVER_ANSWER_SECONDS overrides the bound, 0 = none).  After OFFSIM_EHIP the request is still posted (seq_in has moved): do
 * not call again on this mailbox -- end the server (synchronise its stream or reset the device), clear the mailbox, start afresh. */
#define OFFSIM_SERVER_GONE 1
#define OFFSIM_SERVER_ANSWER_SECONDS 10.0
int offsim_step_server_call(offsim_step_mailbox *mailbox, const void *p_new, int32_t n_actions, int32_t prob_mode, uint32_t cmd,
                            int32_t reject_mode, uint64_t max_spins, int32_t *out3);

/* Sets the current state of masked rollouts (used after a Python-side accept): cur_slot[r] = slot[r]. */
int offsim_env_set_state(offsim_rollouts *ro, const int32_t *slot, const uint8_t *mask, void *stream);

/* Payload of a batched step / reset for many environments at once (the vectorised form of per_state_rejection.py:85-95: the
 * reference returns action, next observation, reward, done of the served row -- `experience[...][row]`).  One launch:
 *   status != NULL (after offsim_step_batch):  ok[k] = status[k] == OFFSIM_ST_OK;  alive[k] &= ok[k]
 *   status == NULL (after offsim_env_reset):   m = mask ? mask[k] : 1;  ok[k] = m && row[k] >= 0;  if (m) alive[k] = row[k] >= 0
 * and for every column c < n_cols: where ok[k], dst_c[k] = src_c[row[k]] (row_bytes bytes); where not, dst_c[k] is left as it
 * is, or zero-filled if zero_if_not_ok (e.g. `done`).  row are rows of the caller's buffer.  n_cols <= 8. */
typedef struct {
    const void *src; /* [n_rows] items of row_bytes bytes, caller's row order */
    void *dst;       /* [R] items */
    int64_t row_bytes;
    int32_t zero_if_not_ok;
    int32_t reserved;
} offsim_column;
int offsim_vector_gather(const int32_t *row, const int32_t *status, const uint8_t *mask, int32_t R, const offsim_column *cols,
                         int32_t n_cols, uint8_t *alive, void *stream);

/* One driver iteration of a batched evaluator in one launch (VectorPSRS.step_and_reset; the loop of
 * examples/cartpole/psrs_from_expert_heuristic.py:59-80 vectorised over environments): PSRS.step(p_new[r]) as offsim_step_batch
 * (advance = 1), then step_cols gathered from the served caller-buffer row into row r of their destinations (zero_if_not_ok columns
 * are cleared where nothing was served), then -- where the served transition ended its episode -- PSRS.reset (offsim_env_reset) and
 * reset_cols gathered from the initial row (typically the observation, written over the next observation).  alive[r] (optional, in/out)
 * is cleared where the step returned None or the reset found the init queue empty.  out_row / out_status (optional) as offsim_step_batch. */
int offsim_vector_step(const offsim_table *t, offsim_rollouts *ro, const void *p_new, int32_t prob_mode, int32_t reject_mode,
                       const offsim_column *step_cols, int32_t n_step_cols, const offsim_column *reset_cols, int32_t n_reset_cols,
                       uint8_t *alive, int32_t *out_row, int32_t *out_status, void *stream);

/* evalMC_psrs(env, n_episodes, pi, gamma) (psrs.py:241-271) for all rollouts in one launch.
 * pi [n_slots,nA] (row s = policy in state slot s), same dtype rule as p_new.
 * gamma_pow [n_gamma_pow] f64 holds gamma**t as the host computes it (Python float ** int == libm pow); Gs are bit-exact
 *   only for t inside the table.  Beyond it: if the table ends stationary (last two entries equal and 0, +-inf or 1 -- for
 *   |gamma| < 1 the factor is exactly 0 from t ~ 7.4e4 on at gamma = 0.99) the last entry is used, which is again exact;
 *   otherwise the device's own pow().  A table of N+1 entries always suffices (an episode has at most N steps).
 * out_sum_g[r] sum of completed episodes' returns (episode order), out_n_ep[r] their number,
 * out_steps[r] accepted steps, out_cand[r] candidates examined, out_n_len[r] entries of `lengths`
 * (n_ep or n_ep+1, psrs.py:265), out_status[r] why the rollout stopped.
 * Optional (NULL to skip): ep_g [R,ep_cap] f64, ep_len [R,ep_cap+1] i32 per-episode values;
 * trace_row [R,trace_cap] i32 accepted caller-buffer rows, trace_pop [R,trace_cap] u32 candidates per step. */
typedef struct offsim_evalmc_out {
    double *sum_g;
    int64_t *n_ep;
    int64_t *steps;
    int64_t *cand;
    int64_t *n_len;
    int32_t *status;
    double *ep_g;
    int32_t *ep_len;
    int64_t ep_cap;
    int32_t *trace_row;
    uint32_t *trace_pop;
    int64_t trace_cap;
    int64_t *dbg; /* optional [R,4] counters of offsim_eval_mc_keys: window-dry events, digest ties, refill phases,
                     64-draw blocks generated; NULL to skip */
} offsim_evalmc_out;

int offsim_eval_mc(const offsim_table *t, offsim_rollouts *ro, const void *pi, int32_t prob_mode, int32_t reject_mode,
                   double gamma, const double *gamma_pow, int64_t n_gamma_pow, int64_t max_episodes,
                   const offsim_evalmc_out *out, void *stream);

/* PSRS_Exo.step (offsim4rl/evaluators/psrs.py:99-117): endogenous state s and exogenous state x have their own queue
 * families; every candidate pops the head of both, the accept/reject test reads the s-row, the accepted s-row gives
 * (r, s', done) and the accepted x-row gives x'.  ts / rs: table grouped by s and its rollout state (rng, cursors,
 * permutations, cur_slot = s); tx / rx: table grouped by x (only z_next and orig_idx are read) and its cursors,
 * permutations and cur_slot = x.  Build, shuffle and reset each with the ordinary entry points (same seeds).
 * out_row_s / out_row_x: caller-buffer rows of the accepted s- and x-elements (-1 on None / KeyError). */
int offsim_step_exo(const offsim_table *ts, const offsim_table *tx, offsim_rollouts *rs, offsim_rollouts *rx,
                    const void *p_new, int32_t prob_mode, int32_t *out_row_s, int32_t *out_row_x, int32_t *out_status,
                    uint32_t *out_popped, void *stream);

/* Learner-in-the-loop drivers qlearn_psrs / expSARSA_psrs (offsim4rl/evaluators/psrs.py:119-239): evalMC's loop plus,
 * after every accepted step (S, A, R, S'):
 *   OFFSIM_TD_QLEARN   Q[S,A] += alpha * (R + gamma * max_a Q[S',a]            - Q[S,A])    (psrs.py:165-168)
 *   OFFSIM_TD_EXPSARSA Q[S,A] += alpha * (R + gamma * sum_a Q[S',a] pi[S',a]   - Q[S,A])    (psrs.py:223)
 * q [R,n_slots,nA] f64 is read as Q_init and written back; td_err [R,td_cap] (optional) gets the TD errors in step
 * order.  pi [n_slots,nA] f64.  Outputs as offsim_eval_mc. */
#define OFFSIM_TD_NONE 0
#define OFFSIM_TD_QLEARN 1
#define OFFSIM_TD_EXPSARSA 2
/* Behaviour policy of the learner drivers (what reveals p_new before every step, psrs.py:158 / :215):
 *   OFFSIM_BEHAVIOUR_FIXED        the tabular `pi` (expSARSA_psrs; qlearn_psrs with a Q-independent policy such as
 *                                 uniformly_random_policy, agents/tabular.py:7-9);
 *   OFFSIM_BEHAVIOUR_EPS_GREEDY   epsilon_greedy_policy on the rollout's own Q row (agents/tabular.py:24-32): epsilon / nA
 *                                 everywhere, 1 - epsilon + epsilon / nA at the arg max; epsilon = 0 is greedy_policy (:11-16);
 *   OFFSIM_BEHAVIOUR_SOFT_GREEDY  soft_greedy_policy (agents/tabular.py:18-22): uniform over the actions whose Q value is
 *                                 np.isclose (rtol 1e-5, atol 1e-8) to the row's maximum.
 * Ties between maxima (EPS_GREEDY): the reference draws np.random.choice among them (agents/tabular.py:4-5), i.e. one masked-
 * rejection bounded integer from NumPy's GLOBAL MT19937 stream per tie and none without a tie.  tie_mt [R,625] u32 is that
 * stream per rollout -- the 624 state words and the position, as np.random.get_state() returns them -- read, advanced and
 * written back; with tie_mt = NULL the FIRST maximum is taken. */
#define OFFSIM_BEHAVIOUR_FIXED 0
#define OFFSIM_BEHAVIOUR_EPS_GREEDY 1
#define OFFSIM_BEHAVIOUR_SOFT_GREEDY 2
typedef struct offsim_td {
    int32_t mode;
    double alpha;
    double *q;
    double *td_err;
    int64_t td_cap;
    int32_t behaviour; /* OFFSIM_BEHAVIOUR_* */
    double epsilon;    /* OFFSIM_BEHAVIOUR_EPS_GREEDY */
    /* schedules (psrs.py:128-135): alpha(episode) / epsilon(episode) tabulated by the caller for episodes 0 .. n_sched-1 (the
     * last entry serves every later episode); NULL: the constants above */
    const double *alpha_ep;
    const double *epsilon_ep;
    int64_t n_sched;
    /* save_Q (psrs.py:172-173, :227-228): Q [n_slots,nA] after every snap_stride-th step (steps 0, stride, 2 stride, ...),
     * q_snap [R,snap_cap,n_slots,nA] f64; NULL: none */
    double *q_snap;
    int64_t snap_cap;
    int64_t snap_stride;
    uint32_t *tie_mt;  /* see above; NULL: first maximum */
    int32_t *beh_arg;  /* [R,td_cap] optional: the action the behaviour policy put its greedy mass on at every step (EPS_GREEDY) */
} offsim_td;
int offsim_eval_td(const offsim_table *t, offsim_rollouts *ro, const double *pi, int32_t reject_mode, double gamma,
                   const double *gamma_pow, int64_t n_gamma_pow, int64_t max_episodes, const offsim_evalmc_out *out,
                   const offsim_td *td, void *stream);

/* Fast path of evalMC_psrs for a fixed tabular policy (the headline scan).
 * offsim_compile_policy folds psrs.py:53-57 for policy pi [n_slots,nA] f64 into one 64-bit key per grouped row:
 *   key = T << 11 | done << 10 | z_next_slot,  T = floor(2^53 * pi[z][a]/p_log[a]/max_a'(pi[z][a']/p_log[a'])),
 * so that  reject <=> u > threshold <=> (53-bit draw) > T, exactly (NaN or >= 1 thresholds give T = 2^53-1).
 * Needs n_slots <= 1024.  keys_out: [N] uint64, caller-owned.
 * offsim_eval_mc_keys runs the same loop as offsim_eval_mc (OFFSIM_PROB_F64, OFFSIM_REJECT_DEFAULT) from those keys,
 * with per-state candidate windows in LDS; n_slots <= 256, otherwise OFFSIM_EUNSUPPORTED (use offsim_eval_mc). */
int offsim_compile_policy(const offsim_table *t, const double *pi, uint64_t *keys_out, void *stream);
/* Name of the kernel offsim_eval_mc_keys launches for this state count and R rollouts ("" if it would refuse): measurement
 * code labels its roofline with it. */
const char *offsim_eval_mc_keys_kernel(int32_t n_slots, int32_t R);
int offsim_eval_mc_keys(const offsim_table *t, offsim_rollouts *ro, const uint64_t *keys, double gamma,
                        const double *gamma_pow, int64_t n_gamma_pow, int64_t max_episodes, const offsim_evalmc_out *out,
                        void *stream);

/* Faults of asynchronous kernels.  Every wait of one wavefront for another (the shuffle's ring protocol, the scan's chain / helper
 * hand-off) is bounded; a wait that gives up ends its workgroup instead of hanging the stream and raises a bit here:
 *   OFFSIM_FAULT_SHUFFLE  offsim_shuffle_queues[_keys]: the orders that call wrote are invalid
 *   OFFSIM_FAULT_SCAN     offsim_eval_mc_streams: the rollouts concerned also report OFFSIM_ST_PROTOCOL
 * offsim_async_faults() returns the bits raised on the current device since its last call and clears them (>= 0; negative: OFFSIM_E*).
 * The entry points themselves return before their kernels run: synchronise the stream first.  No fault has ever been observed in a
 * product build; tests/test_gpu_round3.py raises one with a -DSHUF_FAULT_INJECT build. */
#define OFFSIM_FAULT_SHUFFLE 1
#define OFFSIM_FAULT_SCAN 2
int offsim_async_faults(void);

/* Self-test of the hardware property the headline scan relies on beyond the ISA manual: the LDS applies the lanes of one
 * ds_add_rtn_u32 that hit the same address in ascending lane order (csrc/scan_rows.hpp takes the queue positions of a tick's
 * accepted candidates that way).  *mismatches (device, int64) receives the number of lane operations that returned anything
 * else over ~6e7 randomised ones: 0 on gfx950.  Asynchronous on `stream` like every other entry point. */
int offsim_selftest_lds_atomic_order(int64_t *mismatches, void *stream);
/* The same property as a runtime guard: 1 when the current device has it, 0 when not (or when OFFSIM_FORCE_LDS_ORDER_MISMATCH=1 is in
 * the environment: tests), negative OFFSIM_E* when the test could not run.  The first call on a device runs a short self-test (~1e6
 * lane operations on a stream of its own, synchronised: < 1 ms) and caches the verdict.  offsim_eval_mc_streams and the chunked shuffle
 * (offsim_shuffle_queues[_keys]_ws with a workspace, format C) call it themselves and return OFFSIM_EUNSUPPORTED on 0 -- never wrong
 * numbers; callers that want to route around it (to offsim_eval_mc_keys on permutations and a reset without workspace, as the Python
 * host mirror does) ask first.  The first call on a device allocates and synchronises: those entry points therefore return OFFSIM_EINVAL,
 * with nothing launched, when their stream is being captured and the verdict is not cached yet -- call this once outside the capture
 * (the Python host mirror does, in _lib.require_device()). */
int offsim_lds_order_ok(void);

/* ---- headline scan on per-rollout candidate streams ------------------------------------------------------------
 * For a fixed tabular policy the scan needs, per candidate, only a 32-bit digest of its compiled key (the top bits of the
 * threshold T, done, z_next) -- and per ACCEPTED candidate the row (for its reward).  Gathering digests through a
 * per-rollout permutation moves a 64-byte sector per 4-byte digest, so the sampler reset can instead lay the queue
 * orders out as two streams per rollout, both indexed like `perm` (seg_off[s] + k = k-th element of state s's queue):
 *   dig [n, N] u32   digest of the candidate at that queue position      -> read sequentially by the scan
 *   loc [n, N] u16   its row inside the state's segment (grouped row - seg_off[s]), low 16 bits (format C: u8, low 8 bits)
 * 4 + 2 bytes per queue position and rollout (format C: 4 + 1) in one of three layouts of the digest (offsim_streams.format):
 *   OFFSIM_STREAMS_A   [T >> 32 : 21 | done : 1 | z_next : 10] = the high dword of the compiled key; loc is the whole local row:
 *                      every state has at most 65536 rows;
 *   OFFSIM_STREAMS_B   [T >> 37 : 16 | hi[6:2] : 5 | done : 1 | hi[1:0] : 2 | z_next : 8], hi = bits 16..22 of the local row:
 *                      states of up to 2^23 rows, at most 255 states (a payload of all ones is not a digest).  (The coarser threshold only widens the band of draws that
 *                      are decided by the exact 53-bit look; results are the same bit for bit.)
 *   OFFSIM_STREAMS_C   [T >> 39 : 14 | hi[8:2] : 7 | done : 1 | hi[1:0] : 2 | z_next : 8], hi = bits 8..16 of the local row, loc = its
 *                      low byte: 5 bytes per queue position for states of up to 2^17 rows, at most 255 states -- a sixth less to keep
 *                      resident and to write per sampler reset (a 12.5 M-row shard x 4096 rollouts: 256 GB instead of 307 GB, one
 *                      resident tile instead of two).  Written by offsim_shuffle_queues_keys_ws only (every chain chunk by chunk).
 * offsim_compile_digests: dig32[g] = the digest of grouped row g in `format`, local-row bits zero (from offsim_compile_policy's keys).
 * offsim_shuffle_queues_keys: PSRS.reset_sampler's shuffles (psrs.py:22-23,29-30; same orders as offsim_shuffle_queues,
 *   bit for bit) written as those streams; init_perm_out as in offsim_shuffle_queues.  States of more than 65536 rows need
 *   format B (OFFSIM_EUNSUPPORTED otherwise): their chains are shuffled in place in dig_out and converted afterwards (or see
 *   offsim_shuffle_queues_keys_ws).
 * offsim_eval_mc_streams: evalMC_psrs (psrs.py:241-271) from the streams; same outputs, bit for bit, as offsim_eval_mc /
 *   offsim_eval_mc_keys on the same orders.  `keys` (offsim_compile_policy) is read only to decide digest ties
 *   exactly.  Strides are in elements; stride 0 = one order shared by all rollouts; loc == NULL = queues in table order
 *   (dig = dig32 itself, stride 0; format A).  ro->perm / perm_stride are ignored; ro->init_perm is used as everywhere else.
 *   n_slots <= 256 (255 for formats B and C); offsim_table.max_seg must be set: <= 65536 for format A, <= 2^23 for format B (OFFSIM_EUNSUPPORTED otherwise). */
#define OFFSIM_STREAMS_A 0
#define OFFSIM_STREAMS_B 1
#define OFFSIM_STREAMS_C 2
typedef struct offsim_streams {
    const uint32_t *dig;
    int64_t dig_stride;
    const void *loc;    /* u16 elements (formats A, B) or u8 (format C) */
    int64_t loc_stride; /* in elements */
    int32_t format; /* OFFSIM_STREAMS_* */
} offsim_streams;
int offsim_compile_digests(const offsim_table *t, const uint64_t *keys, int32_t format, uint32_t *dig32_out, void *stream);
int offsim_shuffle_queues_keys(const offsim_table *t, const uint64_t *seeds, int32_t n_perm, const uint32_t *dig32, int32_t format,
                               uint32_t *dig_out, void *loc_out, uint32_t *init_perm_out, void *stream);
/* The same with a workspace lent by the caller (device memory, 8-byte aligned, contents irrelevant before and after): the states of
 * more than 65536 rows are then shuffled chunk by chunk in LDS with sequential global traffic only (csrc/shuffle_chunk.hpp) instead
 * of in place with a random line per swap.  offsim_shuffle_workspace_bytes(t, n) = the bytes n persistent workgroups use (one per
 * compute unit is the most the call starts; 0 = the table has no such state); a smaller workspace runs fewer workgroups, one that
 * holds none (or NULL) gives offsim_shuffle_queues_keys.  Orders are the same bit for bit.  A message list of the chunked kernel
 * that overflowed (probability ~1e-15 per list) raises OFFSIM_FAULT_SHUFFLE (offsim_async_faults): the call's orders are void. */
int64_t offsim_shuffle_workspace_bytes(const offsim_table *t, int32_t n_workgroups);
/* offsim_shuffle_queues with such a workspace: the same permutations, the chains of more than 65536 rows chunk by chunk on chip. */
int offsim_shuffle_queues_ws(const offsim_table *t, const uint64_t *seeds, int32_t n_perm, uint32_t *perm_out, uint32_t *init_perm_out,
                             void *workspace, int64_t workspace_bytes, void *stream);
int offsim_shuffle_queues_keys_ws(const offsim_table *t, const uint64_t *seeds, int32_t n_perm, const uint32_t *dig32, int32_t format,
                                  uint32_t *dig_out, void *loc_out, uint32_t *init_perm_out, void *workspace, int64_t workspace_bytes,
                                  void *stream);
int offsim_eval_mc_streams(const offsim_table *t, offsim_rollouts *ro, const offsim_streams *sm, const uint64_t *keys,
                           double gamma, const double *gamma_pow, int64_t n_gamma_pow, int64_t max_episodes,
                           const offsim_evalmc_out *out, void *stream);

/* ---- encoders (a10, a11) --------------------------------------------------------------------- */

/* CartpoleBoxEncoder.encode (offsim4rl/encoders/heuristic.py:19-71): obs [N,4] f32 -> z [N] i32 in -1..161 */
int offsim_encode_box(const float *obs, int64_t N, int32_t *out_z, void *stream);

/* HOMEREncoder.encode (offsim4rl/encoders/homer.py:159-168) over EncoderModel.obs_encoder
 * (offsim4rl/encoders/models.py:15-19): z = argmax(W2 leaky_relu(W1 x + b1, 0.01) + b2).
 * x [N,dO] f32 (x_dtype OFFSIM_F32) or f16; W1 [H,dO], b1 [H], W2 [nZ,H], b2 [nZ] f32 (state_dict layout).
 * out_logits may be NULL. */
int offsim_encode_mlp(const void *x, int32_t x_dtype, int64_t N, int32_t dO, const float *W1, const float *b1,
                      int32_t H, const float *W2, const float *b2, int32_t nZ, int32_t *out_z, float *out_logits,
                      void *stream);

#ifdef __cplusplus
}
#endif
#endif /* OFFSIM_H */
