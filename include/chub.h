/* chub.h -- C ABI of the MI355X-native vectorised charging-hub environment runtime (libchub.so).
 *
 * Drop-in boundary.  In the reference the Python host (evcssp_env_cpp/envs/evcssp_manager.py:19,
 * class EvcsspManagerEnv_v6) reaches its native core through the Boost.Python module `pyevstation`
 * (evcssp_env_cpp/envs/lion_cpp20/main.cpp:19-290).  This header is what replaces that module for
 * the reset()/step() hot path: plain pointers and sizes, no Python / torch / Boost types.  Each entry
 * point names the reference interface it stands in for.  Everything behind it runs on the GPU; there
 * is no CPU fallback -- every call fails with CHUB_ERR_HIP when no device is usable.
 *
 * Conventions
 *   - N = n_envs, S = piles[0] + piles[1], D = chub_obs_dim(), A = S + 2.
 *   - all arrays are dense row-major; "host" pointers are ordinary CPU memory owned by the caller,
 *     "device" pointers are HIP device memory on the handle's device, owned by the caller.
 *   - every function returns 0 on success or a negative CHUB_ERR_* code; chub_last_error() gives the
 *     message of the calling thread's last failure.  Nothing is printed (the reference prints and
 *     carries on, CHS.hpp:103,292,825).
 *   - a handle is not thread-safe; use one host thread per handle (the reference is single-threaded
 *     with process-global state, CHS.hpp:23-25).
 */
#ifndef CHUB_H
#define CHUB_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CHUB_VERSION 1

enum {
    CHUB_OK = 0,
    CHUB_ERR_ARG = -1,      /* bad argument (the reference: AssertionError / ValueError, MGR:37,148; AGG:196) */
    CHUB_ERR_DATA = -2,     /* data files missing or malformed (the reference: silent UB, CHS.hpp:102-105,733) */
    CHUB_ERR_HIP = -3,      /* HIP runtime failure / no GPU */
    CHUB_ERR_UNSUPPORTED = -4,
    CHUB_ERR_COMM = -5      /* RCCL failure (multi-GPU gather) */
};

enum { CHUB_FAST = 0, CHUB_SLOW = 1 };

/* RNG modes.  COMPAT reproduces the reference's two process-global streams per environment (glibc
 * rand() TYPE_3 + std::minstd_rand0, CHS.hpp:23-45) in the reference's consumption order; exogenous
 * normals / day indices come from the caller (the reference draws them from numpy / random, REN:25,74).
 * PHILOX is the production mode: counter-based Philox4x32-10, key = seed, counter = (block,
 * site<<16|index, tick, global env id); results do not depend on how envs are sharded over GPUs. */
enum { CHUB_RNG_COMPAT = 0, CHUB_RNG_PHILOX = 1 };

/* Constructor kwargs of EvcsspManagerEnv_v6 (MGR:25-27), same names and meaning.  use_lagrange is
 * ignored by the reference (MGR:126) and has no field.  seed_rand maps to the seeds given at create. */
typedef struct chub_config {
    int32_t station_list[2];      /* piles per station; 0 = station absent from obs but still simulated.  At most 4096 per station (the reference's
                                     constructors take any count, CHS.hpp:1148, 1458): CHUB_ERR_UNSUPPORTED beyond -- see INTEGRATION.md */
    int32_t station_type_list[2]; /* CHUB_FAST / CHUB_SLOW */
    int32_t constant_charging;
    int32_t reserved0;
    double hydro_prod_rate;       /* m^3/h  (None -> 430, HYD:140-143) */
    double hydro_store_vlt;       /* m^3    (None -> 5000, HYD:96) */
    double init_soc;              /* 0.1 .. 1 (HYD:137) */
    double fc_max_power;          /* kW     (None -> 100, HYD:401-404) */
    double fcev_permeate;
    double renew_fluctuate;
    double price_fluctuate;
    double hydro_loss;
} chub_config;

typedef struct chub_env chub_env;

/* Build options that are not constructor kwargs of the reference (all zero = defaults; the library reads no
 * environment variables). */
typedef struct chub_options {
    int32_t slot_kernel;  /* PHILOX steps: 0 = the packed slot kernel wherever the hub shape allows (default),
                             1 = the wave-local slot kernel for every step, 2 = same as 0 (kept for tests that name it).
                             COMPAT resets / steps of handles beyond a handful of envs: 0 / 2 = the split form (the stream walks one ENV per
                             lane, then the slots of both stations in one launch, which also leaves the next step's count of empty slots),
                             1 = one kernel per station with the unit's first lane walking (bit-identical; the parity cross-check) */
    int32_t no_arena;     /* 1: one hipMalloc per array instead of one arena (disables chub_get_state / chub_set_state) */
    int32_t fused_step;   /* PHILOX lock-step steps as ONE launch (slot work + per-env tail + next step's draws per workgroup):
                             0 = for small batches, where the two step kernels are launch-bound (default: up to 768 slot workgroups; hubs of fewer than 8 piles
                             up to 384), 1 = never, 2 = always
                             (hub shapes the packed slot kernel covers, stations of at most 64 piles).  Results are bit-identical.
                             COMPAT handles whose envs all fit one workgroup (the drop-in class: one env) run reset and step as one
                             launch too -- station 0, station 1, tail back to back -- unless this is 1. */
    int32_t tile;         /* workgroup tile of the packed slot kernel: 0 = by working-set size (default; a hub of more than 512 piles takes the
                             second tile whatever its batch), 1 = 256 lanes x 2 slots (state and action rows live in the caches), 2 = 512 lanes
                             x 4 slots (they stream from HBM).  Results are bit-identical. */
    int32_t walk_ahead;   /* COMPAT split step: 0 = lock-step steps of every env walk the streams ahead of their step (default): stations of 8 to 64
                             piles run the slot pass of step i and the stream walks of step i + 1 in ONE launch (the walk two steps ahead of the
                             slots it draws for: it takes the slots that will be empty from the stays alone), other shapes the tails of step i
                             and those walks; the walk writes a shadow of the streams that the slot pass of step i + 1 commits, so a reset that
                             comes instead never sees it.  1 = never (every step walks its own streams first: the parity cross-check).
                             Results are bit-identical. */
    int32_t work_order;   /* PHILOX packed kernels, cache-resident sizes (the small tile, at most 6 M charger slots): 0 = tiles, tail and level workgroups
                             take their work in contiguous eighths per XCD (the dispatcher hands workgroup b to XCD b % 8) (default),
                             1 = the dispatcher's order (workgroup b takes tile b: the A/B and the parity cross-check).  Results are bit-identical. */
    int32_t span_steps;   /* chub_run_steps on a PHILOX handle that runs the one-launch step (fused_step): spans of lock-step steps as ONE launch
                             (k_steps_fused: each workgroup goes from step to step by itself, one workgroup barrier between steps; a span ends
                             at the day's end at the latest): 0 = as many steps as the call and the day allow (default), 1 = never (every step
                             a launch: the parity cross-check), n = at most n steps per launch.  Results are bit-identical.  Of a span's packed
                             outputs the last two blocks remain (the policy is open loop over the span: the call's own action batches). */
    int32_t span_tails;   /* ... and where a span's per-env tails run: 0 = by size (default: up to 256 workgroups -- one per CU -- as 2, beyond as 1), 1 = on the
                             workgroup's last slot wave, behind its slot phases (k_steps_fused), 2 = on a fifth wave of the workgroup, ONE STEP BEHIND
                             the slot waves (k_steps_piped: the slot phases of step s + 1 read nothing the tails of step s write, so the two chains of a
                             step run side by side; hubs of 8 piles and more).  Results are bit-identical. */
} chub_options;

/* telemetry column indices of chub_get_telemetry (names of the reference attributes, MGR:183-297) */
enum {
    CHUB_T_HY_ACT = 0, CHUB_T_HY_FLOW_SPEED, CHUB_T_ALL_POWER_SECOND, CHUB_T_STORE_SOC, CHUB_T_CAPACITY,
    CHUB_T_TOTAL_MASS_NEED, CHUB_T_HY_USE, CHUB_T_NOT_MEET, CHUB_T_FC_POWER, CHUB_T_HY_TO_USE,
    CHUB_T_USED_RENEW, CHUB_T_EV0, CHUB_T_EV1, CHUB_T_HYDROGEN_POWER, CHUB_T_INCOME, CHUB_T_REWARD,
    CHUB_T_RE_PV, CHUB_T_RE_WD, CHUB_T_PRICE_NEXT, CHUB_T_HV_ARRIVE, CHUB_T_HV_LINE, CHUB_T_QUEUE_LEN,
    CHUB_T_PV_DAY, CHUB_T_WD_DAY,
    /* ev_power_list / ev_power_sum after the fuel-cell rescale -- what the incomes and cumulated_draw_ele use (MGR:219-224, 262) --
     * and real_state[1] as the step found it (MGR:234) */
    CHUB_T_EV0_NET, CHUB_T_EV1_NET, CHUB_T_EV_SUM_NET, CHUB_T_PRICE_NOW,
    /* per station what make_state puts into real_state (MGR:364-368) + flow_in_number[-1] (MGR:246); valid after reset too */
    CHUB_T_MIN0, CHUB_T_CHG0, CHUB_T_MAX0, CHUB_T_LINE0, CHUB_T_FLOW0, CHUB_T_MIN1, CHUB_T_CHG1, CHUB_T_MAX1, CHUB_T_LINE1, CHUB_T_FLOW1,
    CHUB_T_COUNT
};

/* ---- lifetime ------------------------------------------------------------------------------
 * Replaces: EvcsspManagerEnv_v6.__init__ (MGR:25-130) = Change_Use_Seed (main.cpp:21) + 2x
 * Fast/SlowChargeStation(int,bool,bool) (main.cpp:188,240; AGG:188-196) + HySystem/HFC/ReNew setup.
 * data_dir holds car_flow_possibility_list_save.csv (parsed with the reference's own float parser,
 * CHS.hpp:138-155) and price_96.f64 / pv_100x96.f64 / wd_150x96.f64.  env_id0 = global index of this
 * handle's first env (shard offset); device = HIP device ordinal.  No reset is performed. */
int chub_create(const chub_config *cfg, const char *data_dir, int64_t n_envs, int64_t env_id0, int device,
                uint64_t seed, int rng_mode, chub_env **out);
int chub_create_ex(const chub_config *cfg, const char *data_dir, int64_t n_envs, int64_t env_id0, int device,
                   uint64_t seed, int rng_mode, const chub_options *opt /* NULL = defaults */, chub_env **out);
int chub_destroy(chub_env *env);

int chub_obs_dim(const chub_env *env);  /* 2 + 4*(#stations with piles>0) + 3 (MGR:74-104) */
int chub_act_dim(const chub_env *env);  /* S + 2 (MGR:108-113) */
int64_t chub_num_envs(const chub_env *env);
int chub_clock(const chub_env *env);    /* 0..95: the slot of day, shared by all envs while they run in lock-step (env 0's otherwise) */
int chub_uses_packed_kernel(const chub_env *env); /* 1: PHILOX steps of this handle run k_slot_packed (the production kernel) */
int chub_uses_fused_step(const chub_env *env);    /* 1: its lock-step steps run as one launch (k_step_tailwave / k_step_fused / k_compat_small, small batches) */
int chub_uses_xcd_order(const chub_env *env);     /* 1: its packed kernels' workgroups take their work in XCD-aware order (chub_options.work_order) */

/* ---- hot path ------------------------------------------------------------------------------
 * chub_reset replaces EvcsspManagerEnv_v6.reset (MGR:304-316 -> AGG:157-175 evs_reset main.cpp:199,251,
 * HYD:197-208, REN:51-53).  chub_step replaces EvcsspManagerEnv_v6.step (MGR:136-302 -> AGG:116-155
 * evs_step(Vector_float) main.cpp:198,250; hv_car_number_wrt_poisson / mk_soc main.cpp:151,163).
 *   actions  [N,A] f32 in [-1,1]; pile bit = ((a+1)/2 >= 0.5) in f32; tail is swapped as in MGR:395-403
 *   exo_days [N,2] i32 (pv_day, wd_day) -- required in COMPAT mode, ignored (may be NULL) in PHILOX
 *   exo_z    [N,3] f64 standard normals for the (pv, wd, price) OU updates (REN:71-76) -- COMPAT only
 *   obs      [N,D] f32, reward [N] f32, done [N] u8
 * Host-pointer forms copy in/out around the device-pointer forms. */
int chub_reset(chub_env *env, const int32_t *exo_days, const double *exo_z, float *obs);
int chub_step(chub_env *env, const float *actions, const double *exo_z, float *obs, float *reward, uint8_t *done);
/* The handle's pinned action buffer [N,A] f32: a host that writes its actions there and passes this pointer to chub_step
 * saves the CPU copy into pinned memory (any other host pointer works too). */
int chub_host_actions(chub_env *env, float **out);

/* Packed-action form of chub_step for hosts that sit behind PCIe: all that action_to_real (MGR:384-393) keeps of an action row is
 * one bit per pile ((a + 1) / 2 >= 0.5) and the two tail floats, so that is what travels: pile_bits [N, ceil(S / 64)] u64 (bit b of
 * word w = hub slot 64 w + b: station 0's piles first, as in an action row; 1 = on) and tail [N, 2] f32 = the last two entries
 * of the action row, unchanged.  16 bytes per env for hubs of up to 64 piles instead of 4 (S + 2).  Results are bit for bit those of
 * chub_step on any action rows with the same bits and tail.  chub_host_bits: the handle's pinned staging for both arrays (fill in
 * place and pass these pointers to save a staging copy).  The _device form takes device pointers on `stream`: on the packed slot
 * kernel (PHILOX handles; both launch forms, per-env clocks, recordable into graphs) the step reads the bits themselves -- 8 bytes
 * per env and word of action input instead of a row of floats, and the tail kernel takes the two tail actions from d_tail; on the
 * other kernels the bits are first expanded into action rows on the device. */
int chub_step_bits(chub_env *env, const uint64_t *pile_bits, const float *tail, const double *exo_z, float *obs, float *reward,
                   uint8_t *done);
int chub_host_bits(chub_env *env, uint64_t **pile_bits_out, float **tail_out);
int chub_step_bits_device(chub_env *env, const uint64_t *d_pile_bits, const float *d_tail, const double *d_exo_z, float *d_obs,
                          float *d_reward, uint8_t *d_done, void *stream);
/* ... with the outputs of chub_step_device_packed: one [N, D + 2] f32 block (obs, reward, done as 0 / 1) */
int chub_step_bits_device_packed(chub_env *env, const uint64_t *d_pile_bits, const float *d_tail, const double *d_exo_z, float *d_packed,
                                 void *stream);

/* Same, all pointers device memory, enqueued on `stream` (a hipStream_t, NULL = default stream);
 * returns after enqueueing.  This is the form the multi-GPU host and bench.py use. */
int chub_reset_device(chub_env *env, const int32_t *d_exo_days, const double *d_exo_z, float *d_obs, void *stream);
int chub_step_device(chub_env *env, const float *d_actions, const double *d_exo_z, float *d_obs, float *d_reward,
                     uint8_t *d_done, void *stream);

/* Per-env clocks.  Every reference env is its own object with its own clock: any one of them can be reset, or stepped,
 * while the others are not (MGR:137-140, 271-273, 299, 304-316).  These entry points take a host array mask[n_envs]
 * (non-zero = the env takes part) next to full-size [n_envs, ...] arrays of which only the rows of the named envs are
 * read and written (exo_days / exo_z as for chub_reset / chub_step: COMPAT handles only).  The first such call makes the clock two bytes of per-env device state; a call is
 * still ONE launch (one Philox tick) in which every env it names runs on its own clock, and lock-step use -- chub_reset /
 * chub_step on everybody -- costs what it did.  chub_reset of everybody brings all envs back onto one clock.
 * chub_env_clocks: slot of day of every env and (tick_out may be NULL) the Philox tick of its last launch;
 * chub_clock_groups: number of distinct clocks right now.  (Both read the device: they synchronise.) */
int chub_reset_envs(chub_env *env, const uint8_t *mask, const int32_t *exo_days, const double *exo_z, float *obs);
int chub_step_envs(chub_env *env, const uint8_t *mask, const float *actions, const double *exo_z, float *obs, float *reward,
                   uint8_t *done);
int chub_reset_envs_device(chub_env *env, const uint8_t *mask /* host */, const int32_t *d_exo_days, const double *d_exo_z, float *d_obs,
                           void *stream);
int chub_step_envs_device(chub_env *env, const uint8_t *mask /* host */, const float *d_actions, const double *d_exo_z, float *d_obs,
                          float *d_reward, uint8_t *d_done, void *stream);
int chub_env_clocks(chub_env *env, int32_t *t_out, uint32_t *tick_out);
int chub_clock_groups(chub_env *env);

/* Scalar-load control mode: replaces EvcsspManagerEnv-level use of Fast/SlowChargeStation.evs_step(float)
 * (CHS.hpp:1169-1186 / 1480-1497, bound at main.cpp:196-197,248-249): one kW target per station instead of one bit per
 * pile; the piles are switched on in urgency order until the (clamped) target is met.  Same array layout as chub_step,
 * with actions[i][0] = load of station 0 and actions[i][station_list[0]] = load of station 1 (kW, not normalised; the
 * other pile entries are ignored; a station without piles has no load entry); the two tail entries keep their meaning. */
int chub_step_load(chub_env *env, const float *actions, const double *exo_z, float *obs, float *reward, uint8_t *done);
int chub_step_load_device(chub_env *env, const float *d_actions, const double *d_exo_z, float *d_obs, float *d_reward,
                          uint8_t *d_done, void *stream);
/* ... and on a subset of the envs (mask as for chub_step_envs: every reference station takes evs_step(float) on its own) */
int chub_step_load_envs(chub_env *env, const uint8_t *mask, const float *actions, const double *exo_z, float *obs, float *reward,
                        uint8_t *done);
int chub_step_load_envs_device(chub_env *env, const uint8_t *mask /* host */, const float *d_actions, const double *d_exo_z, float *d_obs,
                               float *d_reward, uint8_t *d_done, void *stream);

/* Packed form for the multi-GPU gather: one [N, D+2] f32 buffer, row = obs[D], reward, done (0.0 / 1.0), so that
 * a shard's whole step output travels in a single RCCL gather. */
int chub_step_device_packed(chub_env *env, const float *d_actions, const double *d_exo_z, float *d_packed, void *stream);

/* Random policy on device: fills d_actions [N,A] with i.i.d. uniform(-1,1) f32 from Philox key
 * `key`, counter (j, 0, batch, global env id).  (test/env_test.py drives the reference with a fixed
 * policy; RL trainers supply their own.) */
int chub_random_actions_device(chub_env *env, uint64_t key, uint32_t batch, float *d_actions, void *stream);

int chub_sync(chub_env *env);

/* ---- multi-GPU: the one collective of the path -----------------------------------------------------------------
 * No counterpart in the reference (it has nothing distributed): environments are independent, so the
 * path shards by contiguous ranges of the global env index -- one process per GPU, handle i created with
 * env_id0 = i * n_local (the Philox streams are keyed by the GLOBAL env id, so results do not depend on the sharding) --
 * and per step each shard's packed output [n_local, D+2] f32 (chub_step_device_packed) travels to rank 0 in ONE grouped
 * ncclSend / ncclRecv over the direct xGMI links, enqueued on the same HIP stream as the step kernels: no host wait,
 * capturable in a hipGraph.  RCCL is loaded on first use (dlopen); a process that never makes a communicator never maps it.
 *   chub_comm_unique_id: rank 0 makes the 128-byte RCCL id; the host passes it to the other ranks (file, pipe, socket).
 *   chub_comm_create:    every rank, same id (ncclCommInitRank on `device`).
 *   chub_comm_gather:    every rank: `bytes` from d_send to rank 0's d_recv + rank * bytes (d_recv ignored elsewhere).  IN PLACE on rank 0:
 *                        with d_send == d_recv its own block already lies where it belongs and rank 0 neither sends to nor receives from
 *                        itself (on a world of one nothing is enqueued at all).
 *   chub_step_gather:    the multi-GPU step: chub_step_device_packed + chub_comm_gather of the packed block on one stream.  Rank 0's step
 *                        kernels write its block STRAIGHT INTO d_gathered (rows 0 .. n_local - 1: the in-place form above -- no copy of
 *                        the root's own block); d_packed is not touched on rank 0 and may be NULL there.  Every other rank steps into
 *                        d_packed and sends it; d_gathered is ignored there.
 *   chub_comm_max_f64 / chub_comm_barrier: max over ranks of one host double / rendezvous (both synchronise `stream`);
 *                        what bench.py brackets its timed region with.
 *   chub_comm_gather_timed: the same gather `reps` times back to back between two HIP events on `stream`: microseconds per gather
 *                        (the per-phase split bench.py prints at N > 1; every rank calls it; synchronises).
 *   chub_comm_world:     the communicator's size as RCCL reports it (ncclCommCount).
 *   chub_comm_ranks_seen: all-reduce sum of one 1 per rank: the number of processes RCCL actually moved data between.
 *   chub_device_info:    out[4] = PCI domain, bus, device of HIP device `device` and its compute-unit count (which physical GPU a
 *                        rank sits on; bench.py lists it per rank). */
typedef struct chub_comm chub_comm;
int chub_comm_unique_id(void *id128);
int chub_comm_create(const void *id128, int world, int rank, int device, chub_comm **out);
int chub_comm_destroy(chub_comm *comm);
int chub_comm_world(const chub_comm *comm);
int chub_comm_rank(const chub_comm *comm);
int chub_comm_gather(chub_comm *comm, const void *d_send, void *d_recv, int64_t bytes, void *stream);
int chub_comm_gather_timed(chub_comm *comm, const void *d_send, void *d_recv, int64_t bytes, void *stream, int reps, double *us_per_gather);
int chub_comm_max_f64(chub_comm *comm, double *value, void *stream);
int chub_comm_barrier(chub_comm *comm, void *stream);
int chub_comm_ranks_seen(chub_comm *comm, int *out, void *stream);
/* Overlapped gathers (off by default).  chub_comm_set_overlap(comm, 1): chub_comm_gather / chub_step_gather put the collective on a
 * stream of the communicator's own, behind an event of the caller's stream, and the caller's stream goes on at once: the gather of step
 * k runs beside the kernels of step k + 1.  Inside a hipGraph capture (chub_graph_begin .. chub_graph_end) the events are graph edges --
 * no host cost per replay; call by call they cost the host two event calls per step (measured slower than the serial form).
 *   chub_comm_gather_begin: before enqueueing work that overwrites a send buffer: `stream` waits for the gather that last read it.
 *                           Announcing a buffer here is what makes its gathers overlapped ones (two gathers may be out at a time: the packed
 *                           step output is double-buffered; chub_step_gather announces its block -- a third buffer waits, on `stream`, for
 *                           the older of the two; a buffer whose gather has been joined gives its slot up); chub_comm_gather of a buffer
 *                           nobody announced goes out on the caller's stream behind every gather still out
 *                           Join (chub_comm_join) BEFORE chub_graph_begin: a capture must not wait on an event recorded outside it.  The
 *                           overlapped form has run on a world of one only; it stays off by default (bench.py --overlap-gather) until a
 *                           run with more than one rank has verified it
 *   chub_comm_join:         `stream` waits for every gather still out: before the gathered blocks are consumed, before a host
 *                           synchronisation that is meant to cover them (chub_graph_end, chub_comm_max_f64 / barrier / ranks_seen /
 *                           gather_timed call it themselves) */
int chub_comm_set_overlap(chub_comm *comm, int enabled);
int chub_comm_gather_begin(chub_comm *comm, const void *d_send, void *stream);
int chub_comm_join(chub_comm *comm, void *stream);
int chub_device_info(int device, int32_t *out4);
int chub_step_gather(chub_env *env, chub_comm *comm, const float *d_actions, float *d_packed, float *d_gathered, void *stream);

/* A run of n_steps steps issued from C, starting at step index first_step: before every step whose index is a multiple of 96 a
 * chub_reset_device (into d_reset_obs), then chub_step_gather (comm != NULL; d_gathered2 may be NULL off rank 0) or
 * chub_step_device_packed, with actions d_action_batches[i % n_batches] and outputs d_packed2[i & 1] / d_gathered2[i & 1].  Exactly
 * the RESULTS of the calls a host loop would make (PHILOX handles); returns after enqueueing.  Without a communicator, on a handle that
 * runs the one-launch step (chub_uses_fused_step: small batches) and with at most 8 action batches, consecutive lock-step steps go out as
 * ONE launch per span (k_steps_piped / k_steps_fused; chub_options.span_steps, span_tails) -- a span ends where the call does, at a reset and where the handle's clock
 * wraps; only the span's last two packed blocks exist afterwards, as after the same steps issued one by one.  Not under the per-kernel
 * profiler, with telemetry on, with a tape loaded or on per-env clocks: every step is then a launch of its own. */
int chub_run_steps(chub_env *env, chub_comm *comm, const float *const *d_action_batches, int n_batches, float *const *d_packed2,
                   float *const *d_gathered2, float *d_reset_obs, int64_t first_step, int64_t n_steps, void *stream);

/* Per-kernel timing of the step: between chub_profile_begin and chub_profile_end every `every`-th step (up to
 * max_steps samples) records HIP events on the launch stream around the slot kernel and the env kernel; _end
 * synchronises and returns the summed durations in milliseconds and the number of steps sampled. */
int chub_profile_begin(chub_env *env, int max_steps, int every);
int chub_profile_end(chub_env *env, double *slot_ms_sum, double *env_ms_sum, int *n_steps);

/* ---- introspection (parity tests, `re_*` telemetry, show_situation MGR:412-414) ---------------
 * chub_get_slots: per env, per station k, field-major [9][piles[k]]: car, charge, emergency, power, soc,
 *   init_soc, target_soc (Station::situation, CHS.hpp:204-231), stay_time, already_stay_time
 *   (CHS.hpp:245-246); out is [N][9*S] f32 with station 0's block first.  Empty slots report the pile
 *   defaults of the reference (-1 for the two counters).
 * chub_get_station_scalars: [N][2][8] f64 = min_power, charge_power, max_power, car_number, line,
 *   flow_in_number[-1], station_time_hole, transformer_limit (AGG:198-218, MGR:368).
 * chub_get_telemetry: [N][CHUB_T_COUNT] f64 of the last step.
 * chub_get_obs_f64 / chub_get_reward_f64: last observation / reward before the f32 narrowing. */
int chub_get_slots(chub_env *env, float *out);
int chub_get_station_scalars(chub_env *env, double *out);
int chub_get_telemetry(chub_env *env, double *out);
int chub_get_obs_f64(chub_env *env, double *out);
int chub_get_reward_f64(chub_env *env, double *out);
int chub_set_telemetry(chub_env *env, int enabled); /* off by default: the hot path then skips those stores */
/* Handles of a few envs (action rows of at most 16 KB in all: the drop-in class runs ONE env) keep the telemetry block in pinned host
 * memory that the device writes directly (no copy back): *telem [CHUB_T_COUNT][N] (column-
 * major: one row per CHUB_T_* index), *obs64 [N][D], *reward64 [N], all f64, owned by the handle, valid while telemetry stays on.
 * Larger handles keep the block in device memory (posted PCIe writes of 27 MB per step at 65 536 envs would stall the tail kernel):
 * CHUB_ERR_UNSUPPORTED here, chub_get_telemetry / chub_get_obs_f64 / chub_get_reward_f64 copy.
 * Contents are current once the call that produced them has completed (any host-pointer entry point returns completed; after a
 * device-pointer call: chub_sync / a stream synchronise).  This is how EvcsspManagerEnv_v6.step() reads everything the reference
 * class exposes after a step (MGR:183-297, 364-372) without one device read. */
int chub_telemetry_host(chub_env *env, double **telem, double **obs64, double **reward64);

/* The FCEV waiting list is unbounded as in the reference (HYD:264-265): the entries a list that still gets served can
 * hold are kept one by one, and once no prefix of it fits into 15 minutes any more (HYD:270-276: nobody is served again
 * until reset and the list only grows) its entries are folded into their count and running sums, which is all the
 * reference ever reads of them again.  chub_fcev_stuck_count: number of envs whose forecourt is currently in that state. */
int chub_fcev_stuck_count(chub_env *env, int64_t *out);

/* COMPAT streams: seeds [N][2] u32 = (srand seed, e.seed()) per env, i.e. what Change_Use_Seed /
 * srand / e.seed would install (CHS.hpp:25-44). */
int chub_set_rng_compat_seeds(chub_env *env, const uint32_t *seeds);
/* Raw COMPAT stream state per env, [N][33] u32: the 31 words of glibc's TYPE_3 ring, the index of its
 * front pointer (rear = front - 3 mod 31), and the minstd_rand0 word -- to continue streams mid-sequence. */
int chub_set_rng_compat_state(chub_env *env, const uint32_t *state);
int chub_get_rng_compat_state(chub_env *env, uint32_t *state);
/* COMPAT: what the reference's constructor does with the two streams before its first reset() (MGR:25-119): one
 * evs_reset per station constructor (CHS.hpp:1152,1462) and the 101-step electrolyser sweep with live FCEV arrivals
 * (HYD:154-157), which also yields each env's hy_power_speed_list.  Call once after chub_create /
 * chub_set_rng_compat_seeds, then chub_reset (= MGR:120). */
int chub_compat_replay_constructor(chub_env *env);
/* Persistent OU states (never reset by the reference, MGR:304-316): [N][3] f64 pv, wd, price. */
int chub_set_ou_state(chub_env *env, const double *ou);

/* ---- hipGraph capture: a launch-bound loop as one submission ---------------------------------------------------------------
 * Between chub_graph_begin and chub_graph_end the device-pointer entry points (chub_reset_device, chub_step_device*,
 * chub_step_gather, chub_random_actions_device) called with `stream` are recorded, not run; chub_graph_launch replays the
 * recording.  What a replay repeats verbatim: clocks, buffers and the order of calls -- so record whole episodes (reset + 96
 * steps), an even number of calls (the state-independent draws are double-buffered), and replay only when the handle is back at
 * the clock the capture started from (slot of day, steps since the last reset mod 4: chub_graph_launch refuses otherwise).
 * What it does not repeat: the random streams -- every replay (and every call issued one by one between replays) moves the
 * Philox tick on, so a replayed episode is a new episode.  PHILOX handles only.
 * `stream` is a created stream (chub_stream_create or the caller's own), not the default stream. */
typedef struct chub_graph chub_graph;
int chub_graph_begin(chub_env *env, void *stream);
int chub_graph_end(chub_env *env, void *stream, chub_graph **out);
int chub_graph_launch(chub_graph *graph, void *stream);
int chub_graph_destroy(chub_graph *graph);

/* ---- device buffers and streams for hosts without a GPU array library (the reference-shaped Python host is ctypes + numpy):
 * plain hipMalloc / hipFree / hipMemcpy (synchronising) / hipStream* on `device`. */
int chub_malloc_device(int device, int64_t bytes, void **out);
int chub_free_device(int device, void *d_ptr);
int chub_copy_to_host(int device, void *dst, const void *d_src, int64_t bytes, void *stream);
int chub_copy_to_device(int device, void *d_dst, const void *src, int64_t bytes, void *stream);
int chub_alloc_host(int device, int64_t bytes, void **out);  /* pinned host memory: the DMA engines reach it directly */
int chub_free_host(int device, void *ptr);
int chub_stream_create(int device, void **out);
int chub_stream_destroy(int device, void *stream);
int chub_stream_sync(int device, void *stream);

/* ---- tape mode: a parity instrument for the production (PHILOX) kernels ------------------------------------------------
 * The production streams are this build's own definition, so the kernels that run them cannot be compared with the
 * reference's recorded trajectories seed for seed.  Tape mode closes that gap: the caller supplies what the streams
 * would have drawn -- per (station, env) unit the step's packed station-level decisions (64 bits: bits 0-9 one renege-pass bit per
 * queue position, 10-13 arrivals n <= 9, 14 + 4 l (l = 0..10) how many of the n arrivals stay if the queue holds l cars after the
 * renege pass; k_draw_levels decodes the word against the unit's live queue, dk_make in chub_kernels.hip) and per admitted car its
 * arrival SoC, target level and extra stay -- and the SAME packed slot kernel replays them.  Arbitrary arrival SoCs are registered as
 * classes of the class table first.  tests/test_gpu_tape.py replays every reference fixture this way, evs_reset included.
 *   chub_tape_register_soc: soc[count] -> class_ids[count].  The caller's arrival SoCs take the place of the handle's own 2048
 *                          classes, first come first row (the slot state has 11 bits for the class), so a handle that registers
 *                          any is a tape handle from then on: chub_reset / chub_step and their device / masked / bits forms return
 *                          CHUB_ERR_ARG on it (cars admitted by the build's own draws would be given the caller's SoCs).  An SoC outside
 *                          0 .. 100, or one whose stay could exceed the state word's 31 slots, is refused.  chub_tape_clear_soc starts
 *                          over at class 0 (between episodes): the next call must be the chub_reset_tape that wipes every slot --
 *                          chub_step_tape returns CHUB_ERR_ARG until then.
 *   chub_set_slots:        rows [N][S][6] i32 in hub order (station 0's slots first): class (-1 = empty), target level,
 *                          stay_time (<= 31), already_stay_time, car_steps taken, charging flag.
 *   chub_set_station_queue: line [N][2] i32 (Station::line).
 *   chub_step_tape:        one step; pk_tape [2][N] u64, car_tape [N][S][2] u32 in hub order = class, level | late << 16
 *                          (read only for slots that admit a car this step).  Host pointers. */
int chub_tape_register_soc(chub_env *env, const float *soc, int32_t count, uint32_t *class_ids);
int chub_tape_clear_soc(chub_env *env);
int chub_set_slots(chub_env *env, const int32_t *rows);
int chub_set_station_queue(chub_env *env, const int32_t *line);
int chub_step_tape(chub_env *env, const float *actions, const uint64_t *pk_tape, const uint32_t *car_tape, float *obs,
                   float *reward, uint8_t *done);
/*   chub_reset_tape:       one reset of every env, evs_reset (CHS.hpp:1209-1231 / 1520-1542) fed the reference's draws: occ_tape [2][N] u32
 *                          = what init_station_car_number and the balk pass of the empty queue came to per unit (arrivals as signed 16
 *                          bits | arrivals that stay << 16: the word k_reset_levels leaves), car_tape as above for the cars admitted. */
int chub_reset_tape(chub_env *env, const uint32_t *occ_tape, const uint32_t *car_tape, float *obs);
/* The WHOLE step / reset from the tape (round 5): the per-env tail of the production step -- k_env<.., PHILOX>, or the tail half of the
 * one-launch step k_step_fused -- takes its variates from the caller as well, where the reference draws them:
 *   exo_z    [N][3] f64  the normals of the PV, wind and price OU processes (np.random.normal inside OU_Noise.sample, renewable.py:71-76,
 *                        evcssp_manager.py:344-361), as the reference's numpy drew them; entries of processes that do not draw are not read
 *   hv_tape  [N][hv_w] u32  the forecourt: word 0 = FCEV arrivals of this step (PoissonNumber.hv_car_number_wrt_poisson, hydro_sys.py:251),
 *                        word 1 + j = arrival j's SoC, f32 bits (CarArriveRandom.mk_soc, hydro_sys.py:259)
 *   exo_days [N][2] i32  a reset's PV / wind days (random.randint inside ReNew.renew_reset, renewable.py:25,51-53)
 * With chub_set_hy_table (hy_power_speed_list as the reference's constructor built it) the observation, reward and every telemetry
 * column of the PRODUCTION kernels can be held to the reference's recorded values directly: tests/test_gpu_tape.py does, on every
 * fixture, in both launch forms (k_slot_packed + k_env, and k_step_fused).  exo_z / hv_tape (exo_days / exo_z) both null: the tail
 * keeps this build's own Philox draws, as chub_step_tape / chub_reset_tape. */
int chub_step_tape_env(chub_env *env, const float *actions, const uint64_t *pk_tape, const uint32_t *car_tape, const double *exo_z,
                       const uint32_t *hv_tape, int32_t hv_w, float *obs, float *reward, uint8_t *done);
int chub_reset_tape_env(chub_env *env, const uint32_t *occ_tape, const uint32_t *car_tape, const int32_t *exo_days, const double *exo_z,
                        float *obs);

/* Snapshot / restore of the whole simulation state (clock, streams, every slot and env variable): checkpoint /
 * resume, planners that branch from a state.  The reference cannot do this (pickling disabled, main.cpp:234; raw
 * back-pointers, CHS.hpp:238).  A snapshot restores only into a handle created with the same arguments. */
int64_t chub_state_size(const chub_env *env);
int chub_get_state(chub_env *env, void *buf, int64_t size);
int chub_set_state(chub_env *env, const void *buf, int64_t size);

/* electrolyser action->power table hy_power_speed_list[102] (HYD:154-157).  The reference builds it at construction
 * with 101 real hy_step()s, i.e. with live random FCEV demand, which matters whenever a tank clamp binds during that
 * sweep.  COMPAT: chub_compat_replay_constructor computes exactly that table per env from the env's streams (until
 * then, and in PHILOX mode, the table is the zero-demand sweep, one per handle: the production mode's definition).
 * chub_get_hy_table_env returns env i's table (PHILOX: the handle's); chub_set_hy_table installs one for every env. */
int chub_get_hy_table(const chub_env *env, double *out102);
int chub_get_hy_table_env(chub_env *env, int64_t env_index, double *out102);
int chub_set_hy_table(chub_env *env, const double *in102);

const char *chub_last_error(void);
int chub_device_count(void);
/* hash of the sources this library was built from (the Python host refuses a library that is older than its sources) */
const char *chub_build_id(void);

#ifdef __cplusplus
}
#endif
#endif /* CHUB_H */
