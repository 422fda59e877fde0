// chub_device.h -- device-side data layout shared by the kernels (chub_kernels.hip) and the host
// runtime (chub_runtime.cpp).  All state lives in HBM as struct-of-arrays:
//
//   per-slot arrays   [station k][env][slot]   (station-major, so that a wave covers one contiguous
//                                               run of one station type: base_k + env*S_k + slot)
//   per-station arrays [k][env]                (queue length, arrivals and the three power sums)
//   per-env arrays     [field][env]            (tank, OU states, exogenous values, price)
//
// Envs run in lock-step (one shared clock), so the clock, the price-noise phase and the Philox tick
// are kernel arguments, not state.
#pragma once

#include <stdint.h>

#include "chub_curves.h"

// Device pointers kept inside structs that live in memory lose their address space (the compiler then emits flat_*
// instead of global_* accesses); spell it out for the device pass.  Same size and layout on both sides.
#if defined(__HIP_DEVICE_COMPILE__)
#define CHUB_G(T) T __attribute__((address_space(1))) *
#else
#define CHUB_G(T) T *
#endif

namespace chub {

constexpr int kMaxLine = 10;     // Station::max_line, CHS.hpp:197
constexpr int kLevels = 1000;    // RandomUtil::uniform_rand has 1000 levels k/999, CHS.hpp:35-44
constexpr int kBalkTab = 1024;   // balk thresholds by queue place, expf(-0.01 m) against the 1000 levels: 0 from m = 691 on, so the last entry stands for every m beyond
constexpr int kMaxPiles = 4096;  // piles per station (k_slot_unit_any: a unit of more than 256 piles is walked in chunks; its scalar-load control ranks the whole unit in LDS)
constexpr int kSocLevels = 2048;   // PHILOX: equiprobable classes of the EV arrival SoC (top 11 bits of a Philox word): the class
                                   // tables of both stations (768 KB) stay resident in every XCD's L2 next to the streamed state
constexpr int kSocLevelShift = 21;
constexpr int kPolarMaxTrials = 32;
#ifndef CHUB_TRACE
#define CHUB_TRACE 0
#endif
#ifndef CHUB_SLOTS_PER_LANE
#define CHUB_SLOTS_PER_LANE 2
#endif
constexpr int kSlotBlock = 256;    // workgroup size of the slot kernels
#ifndef CHUB_PACKED_BLOCK
#define CHUB_PACKED_BLOCK 256
#endif
constexpr int kPackedBlock = CHUB_PACKED_BLOCK;  // workgroup size of the packed slot kernel
constexpr int kSlotsPerLane = CHUB_SLOTS_PER_LANE;  // packed slot kernel: slots per lane (kPackedBlock * kSlotsPerLane virtual lanes per workgroup)
// ... and its second tile, for handles whose slot state + action rows stream from HBM instead of living in the caches: 512 lanes x 4
// slots = 2048 slots per workgroup, 8 KB runs of state and of action rows.  Measured (262 144 envs x [32, 32], float rows):
// 256 x 2: 102.9 us, 512 x 2: 97.3, 1024 x 2: 94.1, 512 x 3: 92.4, 512 x 4: 90.5, 512 x 6: 96.3, 1024 x 4: 131; at 65 536 x [20, 25]
// (cache-resident) 256 x 2 / x 3 / x 4 and 512 x 2 all take 20.8 us, 512 x 4 21.7
#ifndef CHUB_BIG_BLOCK
#define CHUB_BIG_BLOCK 512
#endif
#ifndef CHUB_BIG_SLOTS_PER_LANE
#define CHUB_BIG_SLOTS_PER_LANE 4
#endif
constexpr int kBigBlock = CHUB_BIG_BLOCK, kBigSlotsPerLane = CHUB_BIG_SLOTS_PER_LANE;
#ifndef CHUB_XCD_ANY_TILE
#define CHUB_XCD_ANY_TILE 0  // (tile experiments: 1 = the second tile also takes the XCD-aware order while the streams are cache-resident)
#endif
constexpr int64_t kBigTileSlots = (int64_t) 10 << 20;  // handles of at least this many charger slots take the second tile (chub_options.tile overrides)
constexpr int64_t kXcdOrderSlots = (int64_t) 6 << 20;  // handles of at most this many charger slots (their streams live in the caches) run their
                                                       // step kernels in XCD-aware work order
constexpr int kClsRow = 32;        // PHILOX: entries (power, t_soc) per arrival-SoC class = car_steps a car can take (stay_time <= 27 here)
constexpr int kTelemCount = 38;
constexpr int kCompatSmallBlock = 512;  // k_compat_small: wave 0 walks the envs' streams, ...
constexpr int kCompatSmallWaves0 = 3, kCompatSmallWaves1 = 4;  // ... these many waves hold station 0's / station 1's units
constexpr int kPipedMaxBlocks = 256;   // chub_run_steps's spans with the tails on a wave of their own, a step behind (k_steps_piped), up to this many workgroups --
                                       // one per CU: the slot waves then have nobody else's tails to overlap with.  us per step, tails on their own wave vs on
                                       // the last slot wave: 128 workgroups 4.93 vs 6.13; 192: 4.99 vs 6.12; 256: 5.00 vs 6.18 / 4.85 vs 6.40; 373: 9.40 vs 7.47;
                                       // 384: 9.73 vs 7.25 (two workgroups per CU already run one's tails beside the other's slot phases)
constexpr int kFusedMaxBlocksTailWave = 768;  // ... as k_step_tailwave (hubs of 8 piles and more: the tails on a wave of their own).  us per step as graph replays,
                                              // one launch vs two: 373 workgroups 8.15 vs 9.6; 559: 9.32 vs 10.08; 745: 9.57 vs 10.18; 1118: 14.99 vs 11.41 (up to
                                              // four workgroups of five waves are resident per CU: 1024 in all)
constexpr int kSpanMaxBlocks = 384;           // chub_run_steps's spans of steps in one launch, up to this many workgroups (745: 12.9 us per step against 11.0)
constexpr int kFusedMaxBlocks = 384;   // PHILOX lock-step steps of at most this many slot workgroups run as ONE launch (k_step_fused).  Measured, us per step
                                       // as graph replays, one launch vs two: 8.08 vs 8.82 at 128 workgroups (C2), 8.50 vs 9.30 at 256, 9.42 vs 9.79 at 373,
                                       // 10.79 vs 10.61 at 745, 11.71 vs 11.38 at 1024, 17.9 vs 13.1 at 1490

// Philox draw sites (counter word 1 = site << 16 | index)
enum Site : uint32_t {
    SITE_ARRIVE = 1,  // index = station; word 0 = arrival level, word 1+j = balk level of arrival j
    SITE_INIT = 2,    // index = station; polar trials for the initial-occupancy normal (reset)
    SITE_RENEGE = 3,  // index = station; word w = renege level of queued car w
    SITE_SOC = 5,     // index = hub slot; polar trials for the arrival SoC normal
    SITE_TGT = 6,     // index = hub slot; word 0 = target-SoC level
    SITE_LATE = 7,    // index = hub slot; polar trials for the extra-stay normal
    SITE_HV = 8,      // word 0 = FCEV arrival level
    SITE_HVSOC = 9,   // index = FCEV arrival number; polar trials
    SITE_OU = 10,     // index = 0 pv, 1 wd, 2 price; polar trials
    SITE_DAY = 11     // word 0 -> pv day, word 1 -> wd day (reset)
};

// StationRec::pkd, the integer part of a station record: bits 0-3 Station::line (<= max_line = 10), bits 4-15
// flow_in_number.back() (signed: negative right after the reset of a small fast station, CHS.hpp:1617), bits 16-31 car_number
CHUB_HD uint32_t pkd_make(int line, int flow, int cars) { return (uint32_t) line | (((uint32_t) flow & 0xFFFu) << 4) | ((uint32_t) cars << 16); }
CHUB_HD int pkd_line(uint32_t w) { return (int) (w & 15u); }
CHUB_HD int pkd_flow(uint32_t w) { return ((int) (w << 16)) >> 20; }
CHUB_HD int pkd_cars(uint32_t w) { return (int) (w >> 16); }

struct SlotArrays {          // index = base_k + env*S_k + slot  (station-major)
    // the COMPAT hot record, one 16-byte load and one 16-byte store per slot and step:
    //   .x power      kW at the car's current point of the curve (Station::situation["power"])
    //   .y arrive_soc the car's arrival SoC (Station::situation["init_soc"]): its current SoC = this advanced by the recorded number of car_steps
    //                 (k_replay_soc, on demand).  soc_to_time(target) is NOT kept: it is Tables::ttab[k][level] (checked at create: k_check_ttab)
    //   .z t_soc      soc_to_time(soc)      -- cached, what car_step and calculate_needed both need
    //   .w bits 0-6 stay_time - already_stay_time (0 = empty), bit 7 charging this step, bits 8-14 stay_time,
    //      bits 15-24 target-SoC level l (target = 80 + 20 * l / 999, CHS.hpp:35-44), bits 25-31 car_steps taken since arrival
    CHUB_G(uint32_t) hot;    // COMPAT [NS][4], station-major; PHILOX [N][S0 + S1]: the 4-byte slot state described in
                             // chub_kernels.hip, hub-major (station 0's piles, then station 1's, like an action row)
    CHUB_G(uint8_t) stay8;   // PHILOX [N][S0 + S1]: Station::stay_time of the car in the slot (CHS.hpp:245), written when it is admitted
    CHUB_G(uint32_t) var[2]; // COMPAT, split step [N][S0 + S1][8], hub-major by admission rank, double-buffered by the parity of the step's tick: the r-th car
                             // a unit admits in that step AS add_car MAKES IT (CHS.hpp:864-877), evaluated by the stream walk where its variates are drawn
                             // (walk_rounds / compat_walk_env): ONE 32-byte record -- power, t_target, t_soc (f32 bits), stay | target level << 7 (the hot
                             // record's words x, t_target, z, w), the arrival SoC (what the slot pass puts into the hot record's word y), three words of padding
};

struct StationArrays {       // unit index u = k*N + env
    CHUB_G(uint32_t) rec;        // [2N][4] per-unit record written by k_slot: min_power, charge_power, max_power (f32 bits),
                                 // line | flow_in << 8 | car_number << 16  (Station::line, flow_in_number.back(), car_number)
    CHUB_G(float) tail_act;      // [N][2] PHILOX: the env's two tail actions (electrolyser, fuel cell), copied by the packed slot kernel
                                 // out of the action row it has just read: the tail kernel reads 8 contiguous bytes per env
                                 // instead of one 128-byte line per env of the [N, S+2] action matrix
    CHUB_G(uint8_t) empt;        // COMPAT, split step [2N]: the unit's empty slots once the NEXT step's departures are out: slots with at most one slot
                                 // of stay left (k_slot_split at its end, or k_compat_empties)
    CHUB_G(uint32_t) fa[2];      // COMPAT, split step [2N], by the parity of the step's tick: what the unit's walk came to: flow_in (16 bits, signed) |
                                 // cars admitted << 16 | queue << 24
    // ... and what lets a walk run TWO steps ahead of the slots it draws for (beside the slot pass of the step in between, k_slot_walk2):
    CHUB_G(uint8_t) empt2[2];    // [2N], by the parity of the tick of the slot pass that leaves it: the unit's slots with at most TWO slots of stay left
    CHUB_G(uint8_t) shrt[2];     // [2N], by the parity of the step's tick: how many of the cars the walk admits in that step stay one slot at most
                                 // (their slots are empty again for the next step's admission; make_car's stay, same arithmetic)
    CHUB_G(uint32_t) pk[2];      // PHILOX, double-buffered by tick parity: what a unit's station-level draws of a step come to,
                                 // decoded one launch ahead against the queue the previous step left (dk_make): bits 0-7 queue after
                                 // the renege pass + arrivals that stay = the cars that want a slot, bits 8-15 flow_in.  For a reset:
                                 // the raw initial-occupancy draws of k_reset_levels (arrivals, signed 16 bits | arrivals that stay << 16)
};

struct EnvArrays {           // index = env (or field*N + env)
    CHUB_G(double) cap;          // HyStore.capacity, g
    CHUB_G(double) store_soc;    // HyStore.Store_SOC as last computed by sty_step (stale after the fuel cell, HYD:428)
    CHUB_G(double) ou;           // [3][N] OU states pv, wd, price (REN:56-76), never reset
    CHUB_G(double) price_noise;  // self.price_next noise part (MGR:356)
    // (the exogenous powers and the price produced by the previous make_state, MGR:349-359, are not state: they are functions of
    //  the table rows of the env's slot of day and of the OU states / price noise above, and the tail re-derives them, bit for bit)
    CHUB_G(int16_t) pv_day;
    CHUB_G(int16_t) wd_day;
    CHUB_G(uint8_t) q_len;       // FCEV waiting list: explicit entries (<= HubParams::qcap)
    CHUB_G(uint8_t) hv_line;     // bits 0-6 HyFCEVStation.line, bit 7: the list is stuck (no prefix fits in 15 minutes any more,
                                 // HYD:270-276) and its entries are folded into q_fold / q_fold_cnt
    CHUB_G(double) q_fold;       // [N][2] folded list: sum of its times, sum of its masses, in the reference's left-to-right order
    CHUB_G(uint32_t) q_fold_cnt; // [N]    folded list: number of entries
    CHUB_G(double) hy_env;       // [N][102] COMPAT only: per-env hy_power_speed_list (the reference builds it with live random
                                 // FCEV demand at construction, HYD:154-157, so it depends on the env's streams)
    CHUB_G(uint32_t) drw[2];     // [N][4] PHILOX: a step's state-independent env draws, made one launch ahead (double-buffered by
                                 // tick parity): three OU normals (f32 bits: pv, wind, price) and the first FCEV arrival's SoC (f32 bits)
    CHUB_G(uint8_t) drw_cnt[2];  // [N]    ... and the FCEV arrival count
    CHUB_G(uint32_t) hv_pre[2];  // COMPAT, split step [N][HubParams::hv_w], double-buffered by tick parity: the forecourt's draws of a step as the walk
                                 // made them behind the stations' (hvs_step, HYD:250-260): word 0 arrivals, word 1 + j arrival j's SoC (f32 bits)
    CHUB_G(double) q_time;       // [N][qcap]
    CHUB_G(double) q_mass;
    CHUB_G(double) obs64;        // [N][D]  (telemetry only)
    CHUB_G(double) reward64;     // [N]     (telemetry only)
    CHUB_G(double) telem;        // [kTelemCount][N] (telemetry only)
};

struct CompatRng {           // reference streams, per env: THREE buffers in rotation
    // StepArgs::rng_cur names the buffer that holds the COMMITTED streams (what chub_get_rng_compat_state and snapshots see, what the kernels
    // that walk the streams in place -- one kernel per station, k_compat_small, the constructor sweep -- read and write).  The split step's walk
    // (k_compat_walk, lane = env) reads buffer rng_cur and leaves the streams' state behind its draws in buffer rng_cur + 1 (mod 3), the shadow;
    // once the slot pass of the step those draws belong to has been launched the host moves rng_cur on by one: that IS the commit, nothing
    // is copied (until round 6 station 0's unit of every env copied 33 words).  So a walk may run ahead of its step and a reset that comes
    // instead of that step simply never sees it.  A walk TWO steps ahead (k_slot_walk2, beside the slot pass of the step in between) reads
    // rng_cur + 1 and writes rng_cur + 2.
    CHUB_G(uint32_t) g3[3];      // [N][32]: 31-word glibc TYPE_3 ring + front index in word 31
    CHUB_G(uint32_t) minstd3[3]; // [N]
};

struct Tables {
    CHUB_G(const uint8_t) cnt[2];     // [96][1000] arrivals per station for level k (already scaled + rounded per type)
    CHUB_G(const uint8_t) cnt_hv;     // [96][1000]
    CHUB_G(const uint16_t) thr_renege;  // [kMaxLine]  queued car w stays iff level >= thr
    CHUB_G(const int16_t) thr_balk;     // [kBalkTab]  arrival stays iff level <= thr[line + j]
    CHUB_G(const int16_t) inv_balk;     // [kLevels]   largest m with thr_balk[m] >= level (-1 if none)
    CHUB_G(const double) price;       // [96]
    CHUB_G(const double) pvT;         // [96][100]  (transposed: one row per slot of the day)
    CHUB_G(const double) wdT;         // [96][150]
    CHUB_G(const double) hy_table;    // [102]
    CHUB_G(const float) soc_d_icdf;   // [4097] inverse CDF of clip(N(7,3),1,10)            (PHILOX mode, tools/gen_tables.py)
    CHUB_G(const uint32_t) late_thr;  // [16]   2^32 * CDF of max(0, round(N(2,2)))          (PHILOX mode)
    CHUB_G(const float) normal_icdf;  // [4097] inverse CDF of N(0,1), and
    CHUB_G(const float) normal_tail;  // [4097] its second level for the lowest / highest cell   (PHILOX mode)
    CHUB_G(const double) sin96;       // [96]   sin(2*pi*t/96), the time feature of the observation (MGR:319-320)
    CHUB_G(const float) ttab[2];      // [1000] soc_to_time(target level k) of station k's curve (target = 80 + 20*k/999)
    CHUB_G(const float) ttab2;        // [2][1024] the same two tables in one padded buffer (packed slot kernel: LDS staging)
    CHUB_G(const float) cls[2];       // [kSocLevels + 1][kClsRow][2] PHILOX: per arrival-SoC class of station k's curve:
                                      //   (power, t_soc) after n = 0 .. kClsRow-1 car_steps (entry 0 = what add_car derives, CHS.hpp:864-877)
    CHUB_G(const float) cls_soc0[2];  // [kSocLevels] the class's arrival SoC (introspection); tape mode overwrites rows of both tables
};

struct HubParams {
    int64_t n_envs;
    int64_t env_id0;
    int32_t S[2];
    int32_t type[2];
    int32_t H[2];            // lanes per (env, station) unit: pow2 >= max(1, S_k), <= 64
    int32_t logH[2];
    int32_t U[2];            // lanes per unit of the COMPAT wave-local kernels: max(1, min(S_k, 64)) -- floor(64 / U) units per wave
    int64_t base[2];         // slot-array offset of station k
    int32_t obs_dim, act_dim;
    int32_t constant_charging;
    int32_t rng_mode;
    int32_t telemetry;
    uint32_t key[2];
    CHUB_G(const uint32_t) tick_base;  // PHILOX: added to every launch's host tick (moves only when a captured graph of steps replays)
    CurveConsts cc;
    float transformer_limit[2];
    // hydrogen system constants (HYD)
    double v_h_max, cap_mass, init_soc, hydro_loss, fc_max_power;
    double cells;            // Electrolyser.cell_number
    double v_M;              // 0.082 * 298
    double cpr_w12;          // W_1 + W_2 of the compressor, J/mol
    double renew_fluct1, price_fluct1;  // 1 + fluctuate
    double price_mean, price_std;
    // correctly rounded reciprocals of the tail's run-time constant divisors (div_c in chub_kernels.hip)
    double rc_cells, rc_cap_mass, rc_vm60k, rc_price_std, rc_half_range[2];
    float hv_rate;           // f32(f32(0.3) * f32(permeate))
    int32_t qcap;            // explicit FCEV waiting-list entries per env = max(1, 2 * (max arrivals per step) - 1)
    int32_t hv_w;            // words per env of EnvArrays::hv_pre: 1 + max arrivals per step
    int32_t epb;             // packed slot kernel: whole envs per workgroup = pblock * pslots / (S0 + S1)
    int32_t pblock, pslots;  // packed slot kernel: the handle's tile, (kPackedBlock, kSlotsPerLane) or (kBigBlock, kBigSlotsPerLane)
    int32_t packed;          // PHILOX steps run k_slot_packed (any hub shape of up to 512 piles)
    int32_t compat_split;    // COMPAT resets / steps run empties -> walk (lane = env) -> slots instead of one kernel per station
    int32_t xcd;             // PHILOX packed kernels: tiles, tail and level workgroups in XCD-aware order (xcd_order in chub_kernels.hip)
};

// Everything a kernel needs that does not change from step to step, kept in device memory and passed by pointer
// (as by-value kernel arguments these ~600 bytes were all loaded into SGPRs up front and spilled).
struct DevCtx {
    HubParams hp;
    SlotArrays sl;
    StationArrays st;
    EnvArrays ev;
    CompatRng cr;
    Tables tb;
};

// host-side copies of the device pointers the packed slot kernel takes as kernel arguments (launch_slot)
struct PackedPtrs {
    const EnvArrays *ev;      // host copies of the array tables (launch_env / k_step build TailArgs from them)
    const StationArrays *st;
    uint32_t *hot, *rec;
    uint8_t *stay8;
    uint32_t *pk[2];
    const float *cls[2], *ttab[2], *ttab2;
    uint32_t late8[8];        // the first 8 thresholds of Tables::late_thr, passed to the packed kernel by value
    const Tables *tb;         // host copy of the table pointers
    const double *sin96;      // host copy of Tables::sin96 (the tail gets its slot's value by value)
};

struct StepArgs {
    int32_t t;               // clock of the slot being simulated (0..95)
    uint32_t tick;           // Philox tick (increments on every reset and step)
    int32_t draw_price;      // price_count % 4 == 0 (MGR:354)
    int32_t station_filter;  // -1 both stations in one launch, else only station k (COMPAT: serial streams)
    double price_last;       // env_aggregator.price[-1] seen by this step's make_state
    double price_prev;       // ... and by the previous make_state of a lock-step run: the tariff of the slot before this one
    const float *actions;    // [N][A]
    const uint64_t *act_bits; // or (chub_step_bits on the packed slot kernel): [N][ceil(S / 64)] one bit per pile, actions = null,
    const float *act_tail;    //   and [N][2] the two tail actions
    const double *exo_z;     // [N][3] or null
    const int32_t *exo_days; // [N][2] or null (reset)
    float *obs;              // row i at obs + i*obs_stride (dense: stride D; packed: stride D+2)
    float *reward;           // element i at reward + i*reward_stride
    uint8_t *done;           // [N] u8, or null when done_f32 is used
    float *done_f32;         // packed form: element i at done_f32 + i*reward_stride (0.0 / 1.0)
    int32_t obs_stride, reward_stride;
    int32_t load_mode;       // scalar-load control (evs_step(float)): actions[.][0] / [.][S0] carry one kW target per station
    // tape mode (PHILOX, packed kernel): the step's station-level draws and the per-admission variates come from the caller
    const uint64_t *pk_tape;   // [2N] packed station draws of this step (layout of StationArrays::pk), or null
    const uint32_t *car_tape;  // [NS][2] per slot: arrival-SoC class, target level | extra stay << 16, or null
    // ... and (tail_tape != 0) the per-env tail takes ITS variates from the caller as well: the three exogenous normals from exo_z (f64, as
    // the reference's numpy drew them), a reset's PV / wind days from exo_days, and the forecourt's arrivals from hv_tape
    const uint32_t *hv_tape;   // [N][hv_w] per env: word 0 = FCEV arrivals of this step, word 1 + j = arrival j's SoC (f32 bits; CarArriveRandom.mk_soc, HYD:259)
    int32_t hv_w;
    int32_t tail_tape;
    // per-env clocks (chub_reset_envs / chub_step_envs): the clock is per-env state, [2][N] u16 (bits 0-6 slot of day, bits 8-9
    // price_count & 3): a launch reads [tick & 1] and writes [(tick + 1) & 1] for EVERY env (the envs it does not serve keep
    // theirs).  null: lock-step, the one clock above holds for every env.
    uint16_t *env_clk;
    const uint8_t *env_mask;   // [N] non-zero = the launch serves this env; null: every env
    int32_t env_lo, env_hi;    // the first and the last env the launch serves (0 .. N - 1 without a mask): the grids cover that range
    // The state-independent draws of a PHILOX step (station levels, OU normals, FCEV arrival) are normally made one launch
    // ahead by the previous launch's level blocks.  fresh: this launch makes its own (k_draw_levels in front of the slot
    // kernel, the tail draws inline) and leaves none for the next -- same Philox counters, same values; used whenever the
    // previous launch was not the previous step of every env it serves (per-env clocks, the first lock-step launch after them).
    int32_t fresh;
    // COMPAT, split step: the units' empty-slot counts (StationArrays::empt) are normally left by the previous split pass (k_slot_split
    // knows every slot's remaining stay when it ends); empt_fresh: this launch counts them itself first (k_compat_empties) -- after
    // create, chub_set_state and a pass in another launch form (k_compat_small)
    int32_t empt_fresh;
    // COMPAT, split step: commit_rng: with this slot pass the streams' state the walk left in the shadow buffer becomes the committed one (always, when
    // a walk kernel made this step's draws: the host moves rng_cur on behind the launch); hv_tape / hv_w above then also carry the forecourt's
    // draws of that walk to the tail
    int32_t commit_rng;
    int32_t rng_cur;         // COMPAT: which of CompatRng's three buffers holds the committed streams when this launch starts
    int32_t walked;          // ... and this step's walk has run already (beside the previous step's tails, k_env_walk): the slot launch skips it
    // a walk two steps ahead of the slots (k_slot_walk2: beside the slot pass of the step before its own).  walk_far: it reads the streams
    // as the previous step's walk left them (that step's shadow, being committed in the same launch), the queue from that walk's word
    // and the empty slots as empt2 - cars admitted by that walk + those of them that stay one slot at most.  walk_short: the walk leaves
    // that count (StationArrays::shrt) for the walk behind it
    int32_t walk_far, walk_short;
#if CHUB_TRACE
    // measurement builds only (make KFLAGS=-DCHUB_TRACE=1, tools/experiments/phase_stamps.py): s_memtime stamps at the phase boundaries of
    // the packed slot kernel and of the tail kernel (16 words per workgroup each); null: none taken
    unsigned long long *stamps_slot, *stamps_env;
#endif
};

}  // namespace chub
