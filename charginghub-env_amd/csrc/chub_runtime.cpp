// chub_runtime.cpp -- host side of libchub.so: the C ABI of include/chub.h, table construction at
// create time, HBM state ownership and kernel launches.  No simulation arithmetic of the step runs on
// the host; what does run here is init-time table building (arrival CDF parsing with the reference's
// own parser, level thresholds, the electrolyser action->power sweep, price statistics).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>

#include <algorithm>
#include <string>
#include <vector>

#include "../../include/chub.h"
#include "chub_device.h"

namespace chub {
void launch_slot(bool reset, const HubParams &hp, const DevCtx *ctx, const StepArgs &sa, hipStream_t stream,
                 const PackedPtrs &pp, hipEvent_t ev0, hipEvent_t ev1);
void launch_replay_soc(const HubParams &hp, const DevCtx *ctx, float *d_out, hipStream_t stream);
void launch_check_ttab(const DevCtx *ctx, uint32_t *d_mismatch, hipStream_t stream);
void launch_env(bool reset, const HubParams &hp, const DevCtx *ctx, const StepArgs &sa, hipStream_t stream, hipEvent_t ev0,
                hipEvent_t ev1, const PackedPtrs &pp);
void launch_random_actions(const HubParams &hp, uint64_t key, uint32_t batch, float *d_actions, hipStream_t stream);
void launch_compat_ctor_sweep(const HubParams &hp, const DevCtx *ctx, int rng_cur, hipStream_t stream);
void launch_tick_advance(uint32_t *tick_base, uint32_t by, hipStream_t stream);
void launch_fill_clocks(uint16_t *dst, int64_t n, uint16_t value, hipStream_t stream);
void launch_keep_clocks(uint16_t *dst, const uint16_t *src, int64_t n, hipStream_t stream);
void launch_expand_bits(const HubParams &hp, const uint64_t *d_bits, const float *d_tail, float *d_actions, hipStream_t stream);
void launch_step_fused(const HubParams &hp, const DevCtx *ctx, const StepArgs &sa, hipStream_t stream, const PackedPtrs &pp, hipEvent_t ev0,
                       hipEvent_t ev1);
void launch_compat_small(bool reset, const HubParams &hp, const DevCtx *ctx, const StepArgs &sa, hipStream_t stream, const PackedPtrs &pp);
void launch_steps_fused(const HubParams &hp, const DevCtx *ctx, const StepArgs &sa, hipStream_t stream, const PackedPtrs &pp, int n_steps, int pc0,
                        int64_t first, const float *const *batches, int n_batches, float *const *packed2, bool piped);
bool slot_walk2_covers(const HubParams &hp);
void launch_slot_walk2(const HubParams &hp, const DevCtx *ctx, const StepArgs &sa, const StepArgs &sw, hipStream_t stream, hipEvent_t ev0, hipEvent_t ev1);
void launch_env_walk(const HubParams &hp, const DevCtx *ctx, const StepArgs &sa, const StepArgs &sw, hipStream_t stream, hipEvent_t ev0, hipEvent_t ev1,
                     const PackedPtrs &pp);
}  // namespace chub

using namespace chub;

static thread_local std::string g_err;

static int fail(int code, const std::string &msg) {
    g_err = msg;
    return code;
}
// for the other translation units of the library (chub_comm.cpp); not part of the ABI
extern "C" __attribute__((visibility("hidden"))) int chub_set_last_error_(int code, const char *msg) { return fail(code, msg); }

#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess)                                                                           \
            return fail(CHUB_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));               \
    } while (0)

struct chub_env {
    chub_config cfg;
    HubParams hp;
    SlotArrays sl;
    StationArrays st;
    EnvArrays ev;
    CompatRng cr;
    Tables tb;
    int device;
    bool fused;         // PHILOX lock-step steps of this handle run as ONE launch (k_step_fused): small batches
    bool span_size_ok = false;   // spans of steps in one launch: few enough workgroups (or the one-launch step forced: the measurements)
    bool span_piped = false;     // ... with the tails on a wave of their own, a step behind (k_steps_piped; chub_options.span_tails)
    int span_steps = 0;          // chub_options.span_steps: chub_run_steps's spans of steps in one launch (0: up to a day's rest; 1: never; n: at most n)
    bool no_walk_ahead = false;  // chub_options.walk_ahead = 1: the split COMPAT step never walks ahead (A/B, parity cross-check)
    int rng_cur = 0;           // COMPAT: which of CompatRng's three buffers holds the committed streams (moved on by every commit: chub_device.h)
    uint32_t walked_tick = 0;  // COMPAT split step: the tick whose stream walk has run already, beside the previous step's tails (0: none)
    uint32_t e2_tick = ~0u;    // ... and the tick of the pass after which StationArrays::empt2 holds every unit's count (what a walk two steps ahead needs)
    bool compat_small;  // COMPAT: every env fits one workgroup for both stations: lock-step resets and steps are ONE launch (k_compat_small)
    bool empt_valid;    // COMPAT, split step: StationArrays::empt holds every unit's empty-slot count for the next step (left by the last split pass)
    DevCtx *d_ctx;      // device copy of {hp, sl, st, ev, cr, tb}
    bool ctx_dirty;
    // lock-step clock (MGR:137-140,299; CHS.hpp:1204; AGG:150-151; HYD:192-193 are three copies of it)
    int t;
    int price_count;
    uint32_t tick;             // Philox tick: +1 per launched reset / step (one per clock group and call)
    // per-env clocks (chub_reset_envs / chub_step_envs): once a call names a subset of the envs, the clock is per-env device
    // state (StepArgs::env_clk) and the lock-step clock above is unused -- until everybody is reset again
    bool per_env;
    uint16_t *d_env_clk;            // [2][N], in the arena
    uint8_t *d_mask;                // [2][N], in the arena: the masks of the last two masked calls (the launches in flight read them)
    uint8_t *h_mask;                // [2][N] pinned staging for them
    hipEvent_t mask_done[2];        // recorded behind the launches that read d_mask[i]
    uint32_t mask_seq;
    const uint8_t *cur_mask;        // device copy of the mask of the call in progress
    int64_t mask_lo, mask_hi;       // ... and the first / last env it names
    // a capture on per-env clocks: every masked call gets a device mask of its own (owned by the graph: a replay reads no host
    // memory), and the capture notes which launch of it served each env last (chub_env_clocks after a replay)
    std::vector<void *> cap_masks;
    std::vector<uint32_t> cap_rel;  // [N] launch number (1-based) within the capture of the env's last masked launch, 0 = none
    uint32_t cap_full_rel;          // launch number of the capture's last launch that served every env, 0 = none
    bool graph_per_env0;
    uint32_t graph_full_tick0;
    std::vector<uint32_t> graph_h_tick0;
    std::vector<uint32_t> h_tick;   // [N] tick of the last launch that served the env through a mask (chub_env_clocks)
    uint32_t full_tick;             // tick of the last launch that served every env
    bool predrawn;                  // the last launch served every env: its level blocks left the next step's state-independent draws
    double price[96];
    double hy_table[102];
    std::vector<void *> allocs;
    // all state and tables live in ONE device allocation (carved 256-B aligned): a few large pages instead of ~60
    // scattered small buffers, so the latency-bound kernels do not start with a TLB miss per array
    char *arena;
    size_t arena_size, arena_used;
    // staging for the host-pointer entry points
    float *d_actions, *d_obs, *d_reward;
    uint8_t *d_done;
    double *d_exo_z;
    int32_t *d_exo_days;
    hipStream_t stream;
    // host-pointer entry points: pinned staging + a private stream (lazily made by host_path_init)
    hipStream_t host_stream;
    hipStream_t upload_stream;    // uploads that must stay outside a capture in progress (masks of captured calls)
    float *h_actions, *h_packed;  // pinned: [N][A], [N][D+2]
    uint64_t *h_bits, *d_bits;    // chub_step_bits: [N][ceil(S / 64)] pinned staging and its device copy
    float *h_tail, *d_tail;       //                 [N][2]
    float *d_packed;              // [N][D+2]
    double *h_telem;    // telemetry block in pinned host memory, written by the tail kernel directly: telem [T][N], obs64 [N][D], reward64 [N]
                        // (handles of a few envs: the drop-in class); larger handles keep the block in device memory:
    double *d_telem = nullptr;
    int tape_classes;   // PHILOX tape mode: caller-registered arrival-SoC classes so far
    bool tape_stale = false;  // chub_tape_clear_soc since the last reset: the slots may hold cars of classes that are gone
    bool tape_only = false;  // ... and once there are any, the handle's class rows are the caller's: only tape resets / steps may admit cars
    uint32_t h_late8[8];
    double h_sin96[96];
    std::vector<float> h_cls[2], h_soc0[2], h_ttab[2];  // host copies of the class tables (introspection)
    uint32_t *d_tick_base;     // see HubParams::tick_base
    uint32_t graph_base;       // host mirror of *d_tick_base: ticks covered by the graph replays so far
    uint32_t graph_tick0;      // host state at chub_graph_begin (restored at chub_graph_end: a capture runs nothing)
    int graph_t0, graph_pc0;
    bool graph_predrawn0;
    bool capturing;
    chub_comm *cap_comm = nullptr;  // the communicator whose gathers the capture in progress holds (chub_step_gather)
    const uint64_t *cur_bits;  // set for the duration of chub_step_bits_device on the packed slot kernel: the step reads the
    const float *cur_tail;     //   decision bits and the tail actions themselves (no action rows)
    const uint64_t *tape_pk;   // set for the duration of chub_step_tape
    const uint32_t *tape_car;
    const uint32_t *tape_hv = nullptr;  // ... and, with them, the tail's tape (chub_step_tape_env / chub_reset_tape_env): FCEV arrivals per env
    int tape_hv_w = 0;
    bool tape_tail = false;     // the tail of the call in flight takes its variates from the caller (exo_z, exo_days, tape_hv)
    // optional per-kernel timing with HIP events on the launch stream (chub_profile_*)
    std::vector<hipEvent_t> prof_events;
    size_t prof_used, prof_cap;
    int prof_every, prof_phase;
    bool prof_on;
};

// ------------------------------------------------------------------------------- data loading
// Read2Vector::string_to_float (CHS.hpp:138-155): f32 accumulator, `d *= 0.1` is an f64 product narrowed
// to f32.  Not correctly rounded -- and that changes 79 of the 96 000 arrival-table cells, so it is kept.
static float parse_cell(const char *s, int len) {
    int i = 0;
    float sum = 0;
    while (i < len && s[i] != '.') {
        sum = sum * 10.0f + (float) s[i] - 48.0f;
        ++i;
    }
    ++i;
    float d = 1;
    while (i < len) {
        d = (float) ((double) d * 0.1);
        float t = (float) (s[i] - '0');
        sum = sum + t * d;
        ++i;
    }
    return sum;
}

// Read2Vector::read + file_to_string (CHS.hpp:96-136): keep digits and '.', split on ','
static int load_cdf(const std::string &path, std::vector<float> &cdf) {
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) return fail(CHUB_ERR_DATA, "cannot open " + path);
    std::string cur;
    std::vector<float> row;
    cdf.clear();
    int rows = 0;
    int c;
    auto flush_cell = [&]() {
        if (!cur.empty()) {
            row.push_back(parse_cell(cur.data(), (int) cur.size()));
            cur.clear();
        }
    };
    bool bad = false;
    auto flush_row = [&]() {
        flush_cell();
        if (!row.empty()) {
            if (row.size() != 301) bad = true;
            cdf.insert(cdf.end(), row.begin(), row.end());
            row.clear();
            rows++;
        }
    };
    while ((c = fgetc(f)) != EOF) {
        if ((c >= '0' && c <= '9') || c == '.') cur.push_back((char) c);
        else if (c == ',') flush_cell();
        else if (c == '\n') flush_row();
    }
    flush_row();
    fclose(f);
    if (bad || rows != 96) return fail(CHUB_ERR_DATA, path + ": expected 96 rows x 301 columns");
    return 0;
}

static int load_f64(const std::string &path, double *dst, size_t count) {
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) return fail(CHUB_ERR_DATA, "cannot open " + path);
    size_t got = fread(dst, sizeof(double), count, f);
    int extra = fgetc(f);
    fclose(f);
    if (got != count || extra != EOF) return fail(CHUB_ERR_DATA, path + ": wrong size");
    return 0;
}

// RandomUtil::uniform_rand(0,1) at level k (CHS.hpp:35-44)
static float level_value(int k) {
    float tr = (float) k / 999.0f;
    tr = tr * (1.0f - 0.0f) + 0.0f;
    return tr;
}

// numpy pairwise sum of 96 values (np.mean / np.std, MGR:45-46)
static double np_sum96(const double *a) {
    double r[8];
    for (int j = 0; j < 8; j++) r[j] = a[j];
    for (int i = 8; i < 96; i += 8)
        for (int j = 0; j < 8; j++) r[j] += a[i + j];
    return ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
}

template <typename T>
static int dev_alloc(chub_env *e, T **p, size_t count, bool zero = true) {
    const size_t bytes = ((count ? count : 1) * sizeof(T) + 255) & ~(size_t) 255;
    void *q = nullptr;
    if (e->arena && e->arena_used + bytes <= e->arena_size) {
        q = e->arena + e->arena_used;
        e->arena_used += bytes;
    } else {
        HIP_TRY(hipMalloc(&q, bytes));
        e->allocs.push_back(q);
    }
    if (zero) HIP_TRY(hipMemset(q, 0, bytes));
    *p = (T *) q;
    return 0;
}

template <typename T>
static int dev_upload(chub_env *e, const T **p, const std::vector<T> &v) {
    T *q = nullptr;
    int rc = dev_alloc(e, &q, v.size(), false);
    if (rc) return rc;
    HIP_TRY(hipMemcpy(q, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    *p = q;
    return 0;
}

static int pow2_ge(int v) {
    int p = 1;
    while (p < v) p <<= 1;
    return p;
}

// HySystem.__init__ sweep (HYD:154-158) with zero FCEV demand: electrolyser + compressor power of the
// flow each request level yields while the tank integrates it (see chub_set_hy_table in chub.h).
static void build_hy_table(const HubParams &hp, double *table) {
    double cap = hp.init_soc * hp.cap_mass;
    for (int i = 0; i < 101; i++) {
        double gen_speed = 0.01 * i;
        double must_charge = hp.cap_mass * 0.1 - cap;
        must_charge = must_charge > 0 ? must_charge : 0.0;
        double upper_charge = hp.cap_mass - cap;
        upper_charge = upper_charge > 0 ? upper_charge : 0.0;
        double charge_temp = gen_speed * hp.v_h_max * (15 * 60);
        charge_temp = charge_temp < upper_charge ? charge_temp : upper_charge;
        charge_temp = charge_temp > must_charge ? charge_temp : must_charge;
        double flow = charge_temp / (15 * 60);
        flow = flow < hp.v_h_max ? flow : hp.v_h_max;
        double ele = 0.0;
        if (hp.cells != 0.0) {
            double v_H_mass = flow / hp.cells;
            double v_H_mol = v_H_mass / 2.02;
            double v_H_L = v_H_mol * hp.v_M;
            double v_H = v_H_L * 1000 * 60;
            double temp = v_H * 2 * 96487 / (hp.v_M * 1000 * 60);
            double power = pow(temp, 2) * 0.326 + temp * 1.476;
            ele = hp.cells * power / 1000;
        }
        double cpr = ((flow / 2.02) * hp.cpr_w12 / 0.8) / 1000;
        cap += flow * 15 * 60;
        cap -= cap * hp.hydro_loss;
        table[i] = ele + cpr;
    }
    table[101] = table[100];
}

// PHILOX: one class row = (power, t_soc) of a car that arrived with `soc` after n = 0 .. kClsRow-1 car_steps
// (add_car CHS.hpp:864-877 / 1029-1042 for entry 0, car_step CHS.hpp:900-905 / 1065-1070 from entry to entry), evaluated with
// the curve functions of chub_curves.h on the host
static void build_class_row(bool fast, bool cp, const CurveConsts &cc, float soc, float *row /* [kClsRow][2] */) {
    float t_soc = fast ? fast_soc_to_time(soc, cp) : slow_soc_to_time(soc, cp);
    float power = fast ? fast_time_to_power(t_soc, cp) : slow_time_to_power(t_soc, cp);
    for (int n = 0; n < kClsRow; n++) {
        row[2 * n] = power;
        row[2 * n + 1] = t_soc;
        const float tt = t_soc + 1.0f;
        const float soc_n = fast ? fast_time_to_soc(tt, cp, cc) : slow_time_to_soc(tt, cp, cc);
        power = fast ? fast_time_to_power(tt, cp) : slow_time_to_power(tt, cp);
        t_soc = fast ? fast_soc_to_time(soc_n, cp) : slow_soc_to_time(soc_n, cp);
    }
}

template <typename T>
static int fetch(std::vector<T> &dst, const T *src, size_t count) {
    dst.resize(count);
    HIP_TRY(hipMemcpy(dst.data(), src, count * sizeof(T), hipMemcpyDeviceToHost));
    return 0;
}

// upload {hp, arrays, tables} when something in them changed (create, telemetry toggle)
static PackedPtrs packed_ptrs(const chub_env *e) {
    PackedPtrs p;
    p.ev = &e->ev;
    p.st = &e->st;
    p.hot = (uint32_t *) e->sl.hot;
    p.stay8 = (uint8_t *) e->sl.stay8;
    p.rec = (uint32_t *) e->st.rec;
    p.cls[0] = e->tb.cls[0];
    p.cls[1] = e->tb.cls[1];
    p.ttab[0] = e->tb.ttab[0];
    p.ttab[1] = e->tb.ttab[1];
    p.ttab2 = e->tb.ttab2;
    memcpy(p.late8, e->h_late8, sizeof p.late8);
    p.tb = &e->tb;
    p.sin96 = e->h_sin96;
    p.pk[0] = (uint32_t *) e->st.pk[0];
    p.pk[1] = (uint32_t *) e->st.pk[1];
    return p;
}

static int sync_ctx(chub_env *e, hipStream_t s) {
    if (!e->ctx_dirty) return 0;
    DevCtx h;
    h.hp = e->hp; h.sl = e->sl; h.st = e->st; h.ev = e->ev; h.cr = e->cr; h.tb = e->tb;
    HIP_TRY(hipStreamSynchronize(s));
    HIP_TRY(hipMemcpy(e->d_ctx, &h, sizeof h, hipMemcpyHostToDevice));
    e->ctx_dirty = false;
    return 0;
}

#if CHUB_TRACE
static void *g_stamps_slot = nullptr, *g_stamps_env = nullptr;
extern "C" void chub_debug_stamps(void *slot, void *env) { g_stamps_slot = slot; g_stamps_env = env; }  // measurement builds only
#endif
extern "C" {

const char *chub_last_error(void) { return g_err.c_str(); }

int chub_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int chub_device_info(int device, int32_t *out4) {
    if (!out4) return fail(CHUB_ERR_ARG, "null argument");
    hipDeviceProp_t p;
    HIP_TRY(hipGetDeviceProperties(&p, device));
    out4[0] = p.pciDomainID;
    out4[1] = p.pciBusID;
    out4[2] = p.pciDeviceID;
    out4[3] = p.multiProcessorCount;
    return CHUB_OK;
}

int chub_create(const chub_config *cfg, const char *data_dir, int64_t n_envs, int64_t env_id0, int device,
                uint64_t seed, int rng_mode, chub_env **out) {
    return chub_create_ex(cfg, data_dir, n_envs, env_id0, device, seed, rng_mode, nullptr, out);
}

int chub_create_ex(const chub_config *cfg, const char *data_dir, int64_t n_envs, int64_t env_id0, int device,
                   uint64_t seed, int rng_mode, const chub_options *opt_in, chub_env **out) {
    if (!cfg || !data_dir || !out) return fail(CHUB_ERR_ARG, "null argument");
    chub_options opt;
    memset(&opt, 0, sizeof opt);
    if (opt_in) opt = *opt_in;
    if (opt.slot_kernel < 0 || opt.slot_kernel > 2) return fail(CHUB_ERR_ARG, "chub_options.slot_kernel must be 0, 1 or 2");
    if (opt.fused_step < 0 || opt.fused_step > 2) return fail(CHUB_ERR_ARG, "chub_options.fused_step must be 0, 1 or 2");
    if (opt.tile < 0 || opt.tile > 2) return fail(CHUB_ERR_ARG, "chub_options.tile must be 0, 1 or 2");
    if (opt.walk_ahead < 0 || opt.walk_ahead > 1) return fail(CHUB_ERR_ARG, "chub_options.walk_ahead must be 0 or 1");
    if (opt.work_order < 0 || opt.work_order > 1) return fail(CHUB_ERR_ARG, "chub_options.work_order must be 0 or 1");
    if (opt.span_steps < 0 || opt.span_steps > 96) return fail(CHUB_ERR_ARG, "chub_options.span_steps must be 0 .. 96");
    if (opt.span_tails < 0 || opt.span_tails > 2) return fail(CHUB_ERR_ARG, "chub_options.span_tails must be 0 (by size), 1 (on the last slot wave) or 2 (on a wave of their own)");
    *out = nullptr;
    if (n_envs <= 0) return fail(CHUB_ERR_ARG, "n_envs must be positive");
    if (n_envs * (int64_t) (cfg->station_list[0] + cfg->station_list[1] + 2) >= (int64_t) 1 << 31)
        return fail(CHUB_ERR_UNSUPPORTED, "n_envs * (piles + 2) must stay below 2^31 per handle (32-bit slot indices)");
    if (rng_mode != CHUB_RNG_COMPAT && rng_mode != CHUB_RNG_PHILOX) return fail(CHUB_ERR_ARG, "unknown rng_mode");
    for (int k = 0; k < 2; k++) {
        if (cfg->station_list[k] < 0) return fail(CHUB_ERR_ARG, "station_list entries must be >= 0");
        // the production (PHILOX) kernel lays whole envs over a workgroup's 512 (2048) virtual lanes; the wave-local kernels keep a unit
        // of up to 64 piles inside a wave, give a larger one a workgroup of 256 lanes (k_slot_unit) and walk a unit of more than 256
        // piles in chunks (k_slot_unit_any, whose scalar-load control ranks the whole unit in LDS: kMaxPiles)
        if (cfg->station_list[k] > kMaxPiles) return fail(CHUB_ERR_UNSUPPORTED, "more than 4096 piles per station is not supported");
        if (cfg->station_type_list[k] != CHUB_FAST && cfg->station_type_list[k] != CHUB_SLOW)
            return fail(CHUB_ERR_ARG, "EVS type must be fast or slow");  // AGG:196
    }
    if (cfg->station_list[0] + cfg->station_list[1] < 1)
        return fail(CHUB_ERR_ARG, "A station must have fast pile or slow pile!");  // MGR:336
    if (!(cfg->init_soc >= 0.1 && cfg->init_soc <= 1)) return fail(CHUB_ERR_ARG, "init_soc must be in [0.1, 1]");  // HYD:137
    if (!(cfg->hydro_prod_rate >= 0) || !(cfg->hydro_store_vlt > 0) || !(cfg->fc_max_power >= 0))
        return fail(CHUB_ERR_ARG, "hydrogen system sizes must be non-negative");

    chub_env *e = new chub_env();
    e->cfg = *cfg;
    e->device = device;
    e->t = 0;
    e->price_count = 0;
    e->tick = 0;
    e->per_env = false;
    e->d_env_clk = nullptr;
    e->d_mask = nullptr;
    e->h_mask = nullptr;
    e->mask_done[0] = e->mask_done[1] = nullptr;
    e->mask_seq = 0;
    e->cur_mask = nullptr;
    e->cap_full_rel = 0;
    e->graph_per_env0 = false;
    e->graph_full_tick0 = 0;
    e->full_tick = 0;
    e->predrawn = false;
    e->stream = nullptr;
    e->host_stream = nullptr;
    e->upload_stream = nullptr;
    e->h_actions = e->h_packed = e->d_packed = nullptr;
    e->h_bits = e->d_bits = nullptr;
    e->h_tail = e->d_tail = nullptr;
    e->h_telem = nullptr;
    e->empt_valid = false;
    e->prof_used = e->prof_cap = 0;
    e->prof_on = false;
    e->arena = nullptr;
    e->arena_size = e->arena_used = 0;
    auto bail = [&](int rc) {
        chub_destroy(e);
        return rc;
    };

    // ---- tables from the data files
    std::string dir(data_dir);
    std::vector<float> cdf;
    int rc;
    if ((rc = load_cdf(dir + "/car_flow_possibility_list_save.csv", cdf))) return bail(rc);
    std::vector<double> pv(100 * 96), wd(150 * 96);
    if ((rc = load_f64(dir + "/price_96.f64", e->price, 96))) return bail(rc);
    if ((rc = load_f64(dir + "/pv_100x96.f64", pv.data(), pv.size()))) return bail(rc);
    if ((rc = load_f64(dir + "/wd_150x96.f64", wd.data(), wd.size()))) return bail(rc);

    {   // data files are validated before the device is touched (so bad data reports CHUB_ERR_DATA everywhere)
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
            return bail(fail(CHUB_ERR_HIP, "no HIP device available: libchub has no CPU path"));
        if (device < 0 || device >= ndev) return bail(fail(CHUB_ERR_ARG, "device ordinal out of range"));
        hipError_t he = hipSetDevice(device);
        if (he != hipSuccess) return bail(fail(CHUB_ERR_HIP, std::string("hipSetDevice: ") + hipGetErrorString(he)));
    }

    // FCEV arrivals per step at most (the count table below is built from the same expression): sizes the explicit part
    // of the waiting list
    int hv_max_arrive = 0;
    {
        float hv_pin = (float) 0.3, hv_perm = (float) cfg->fcev_permeate;
        if (hv_perm > 1) hv_perm = (float) 0.01;
        for (int t = 0; t < 96; t++) {
            // the largest arrival index of row t: level 999 (u = 1), or 300 when the row's CDF never reaches it (CHS.hpp:731-743)
            int n = 300;
            for (int j = 0; j < 301; j++)
                if ((double) cdf[t * 301 + j] >= (double) level_value(kLevels - 1)) {
                    n = j;
                    break;
                }
            int c = (int) roundf(hv_pin * hv_perm * (float) n);
            c = c < 0 ? 0 : (c > 255 ? 255 : c);
            hv_max_arrive = c > hv_max_arrive ? c : hv_max_arrive;
        }
    }
    if (hv_max_arrive > 127) return bail(fail(CHUB_ERR_UNSUPPORTED, "more than 127 FCEV arrivals per step"));
    const int qcap = hv_max_arrive > 0 ? 2 * hv_max_arrive - 1 : 1;

    {   // arena: generous upper bound of everything allocated below (telemetry buffers come later, separately)
        const size_t S_tot = (size_t) (cfg->station_list[0] + cfg->station_list[1]);
        const size_t per_env = S_tot * (rng_mode == CHUB_RNG_COMPAT ? 88 : 40) + 1024 + (size_t) qcap * 16 +  // (COMPAT: 16 + 4 + 2 * 32 bytes per slot)
                               (rng_mode == CHUB_RNG_COMPAT ? 102 * 8 + 3 * 33 * 4 + 2 * 4 * (size_t) (1 + hv_max_arrive) + 1024 + 64 : 0);
        const size_t want = (size_t) n_envs * per_env + ((size_t) 8 << 20) +
                            (rng_mode == CHUB_RNG_PHILOX ? 2 * ((size_t) kSocLevels + 2) * (kClsRow * 8 + 4) : 0);
        void *q = nullptr;
        if (!opt.no_arena && hipMalloc(&q, want) == hipSuccess) {
            e->arena = (char *) q;
            e->arena_size = want;
            e->allocs.push_back(q);
        }
    }

    HubParams &hp = e->hp;
    memset(&hp, 0, sizeof hp);
    hp.n_envs = n_envs;
    hp.env_id0 = env_id0;
    int active = 0;
    for (int k = 0; k < 2; k++) {
        hp.S[k] = cfg->station_list[k];
        hp.type[k] = cfg->station_type_list[k];
        hp.H[k] = pow2_ge(hp.S[k] > 0 ? (hp.S[k] < 64 ? hp.S[k] : 64) : 1);  // wave-local kernels: units of at most one wave
        hp.logH[k] = 0;
        while ((1 << hp.logH[k]) < hp.H[k]) hp.logH[k]++;
        hp.U[k] = hp.S[k] > 0 ? (hp.S[k] < 64 ? hp.S[k] : 64) : 1;  // COMPAT wave-local kernels: units of exactly S lanes
        active += hp.S[k] > 0;
        // transformer_limit = constant_power * charge_number in f32 (CHS.hpp:1133-1134, 1443-1444)
        float constant_power = hp.type[k] == CHUB_FAST ? (float) 36.44764034125146 : (float) 5.254973139368931;
        hp.transformer_limit[k] = constant_power * (float) hp.S[k];
    }
    hp.base[0] = 0;
    hp.base[1] = n_envs * hp.S[0];
    hp.obs_dim = 2 + 4 * active + 3;
    hp.act_dim = hp.S[0] + hp.S[1] + 2;
    hp.constant_charging = cfg->constant_charging ? 1 : 0;
    hp.rng_mode = rng_mode;
    hp.telemetry = 0;
    hp.qcap = qcap;
    hp.hv_w = 1 + hv_max_arrive;
    hp.key[0] = (uint32_t) seed;
    hp.key[1] = (uint32_t) (seed >> 32);
    hp.cc = make_curve_consts();
    // hydrogen system (HYD:94-98, 144, 10-24, 57-72)
    hp.cap_mass = (0.089 * (200 / 1)) * (cfg->hydro_store_vlt * 1000);
    hp.v_h_max = 0.089 * cfg->hydro_prod_rate * 1000 / 3600;
    hp.init_soc = cfg->init_soc;
    hp.hydro_loss = cfg->hydro_loss;
    hp.fc_max_power = cfg->fc_max_power;
    hp.v_M = 0.082 * (273 + 25) / 1;
    {
        double v_H_L = (10.0 / 1000) / 60;
        double v_H_mol = v_H_L / hp.v_M;
        double v_H_mass = v_H_mol * 2.02;
        hp.cells = ceil(hp.v_h_max / v_H_mass);
        double alpha = 1.4, R = 0.082, T = 273 + 25, P_in = 1, P_out = 200;
        double P_a = sqrt(P_in * P_out);
        double part1 = alpha / (alpha - 1);
        double part2 = part1 * R * T;
        double part3 = (alpha - 1) / alpha;
        double W_1 = part2 * (-1 + pow(P_a / P_in, part3));
        double W_2 = part2 * (-1 + pow(P_out / P_a, part3));
        hp.cpr_w12 = W_1 + W_2;
    }
    hp.rc_cells = hp.cells != 0.0 ? 1.0 / hp.cells : 0.0;
    hp.rc_cap_mass = 1.0 / hp.cap_mass;
    hp.rc_vm60k = 1.0 / (hp.v_M * 1000 * 60);
    for (int k = 0; k < 2; k++) {
        const double half_range = (double) hp.transformer_limit[k] / 2;
        hp.rc_half_range[k] = half_range != 0.0 ? 1.0 / half_range : 0.0;
    }
    hp.renew_fluct1 = 1 + cfg->renew_fluctuate;
    hp.price_fluct1 = 1 + cfg->price_fluctuate;
    {
        double mean = np_sum96(e->price) / 96;
        double dev[96];
        for (int i = 0; i < 96; i++) {
            double d = fabs(e->price[i] - mean);
            dev[i] = d * d;
        }
        hp.price_mean = mean;
        hp.price_std = sqrt(np_sum96(dev) / 96);
        hp.rc_price_std = 1.0 / hp.price_std;
    }

    // arrival index per (slot of day, level): first j with CDF[t][j] >= u_k, else 300 (CHS.hpp:731-743),
    // then the per-station-type scaling + std::round (CHS.hpp:751-780, HYD:247-251)
    std::vector<uint8_t> cnt[2], cnt_hv(96 * kLevels);
    cnt[0].resize(96 * kLevels);
    cnt[1].resize(96 * kLevels);
    float hv_pin = (float) 0.3, hv_perm = (float) cfg->fcev_permeate;
    if (hv_perm > 1) hv_perm = (float) 0.01;
    for (int t = 0; t < 96; t++) {
        for (int k = 0; k < kLevels; k++) {
            float u = level_value(k);
            int n = 300;
            for (int j = 0; j < 301; j++) {
                if ((double) cdf[t * 301 + j] >= (double) u) {
                    n = j;
                    break;
                }
            }
            for (int s = 0; s < 2; s++) {
                float possible_in = hp.type[s] == CHUB_FAST ? (float) 0.15 : (float) 0.1;
                float permeability = (float) 0.2;
                float po = possible_in * permeability * (float) n;
                int c = (int) roundf(po);
                cnt[s][t * kLevels + k] = (uint8_t) (c < 0 ? 0 : (c > 255 ? 255 : c));
            }
            int c = (int) roundf(hv_pin * hv_perm * (float) n);
            cnt_hv[t * kLevels + k] = (uint8_t) (c < 0 ? 0 : (c > 255 ? 255 : c));
        }
    }
    if (rng_mode == CHUB_RNG_PHILOX) {
        // the packed per-step station draws (draw_station_levels) hold up to 9 arrivals per station and step
        int top = 0;
        for (int sidx = 0; sidx < 2; sidx++)
            for (uint8_t c : cnt[sidx]) top = c > top ? c : top;
        if (top > 9)
            return bail(fail(CHUB_ERR_UNSUPPORTED, "arrival table yields more than 9 EV arrivals per station and step"));
    }
    // level thresholds of the renege / balk tests (CHS.hpp:1286-1303): u > 0.1*logf(w+1), u <= expf(-0.01*m)
    std::vector<uint16_t> thr_renege(kMaxLine);
    for (int w = 0; w < kMaxLine; w++) {
        float leave_possibility = (float) (0.1 * (double) logf((float) (w + 1)));
        int thr = kLevels;
        for (int k = 0; k < kLevels; k++)
            if (level_value(k) > leave_possibility) {
                thr = k;
                break;
            }
        thr_renege[w] = (uint16_t) thr;
    }
    std::vector<int16_t> thr_balk(kBalkTab);
    for (int m = 0; m < kBalkTab; m++) {
        float stay = expf((float) (-(0.01 * m)));
        int thr = -1;
        for (int k = 0; k < kLevels; k++)
            if (level_value(k) <= stay) thr = k;
        thr_balk[m] = (int16_t) thr;
    }
    std::vector<int16_t> inv_balk(kLevels);
    for (int v = 0; v < kLevels; v++) {
        int best = -1;
        for (int m = 0; m < kBalkTab; m++)
            if ((int) thr_balk[m] >= v) best = m;
        inv_balk[v] = (int16_t) best;
    }
    std::vector<double> price_v(e->price, e->price + 96), pvT(96 * 100), wdT(96 * 150);
    for (int d = 0; d < 100; d++)
        for (int t = 0; t < 96; t++) pvT[t * 100 + d] = pv[d * 96 + t];
    for (int d = 0; d < 150; d++)
        for (int t = 0; t < 96; t++) wdT[t * 150 + d] = wd[d * 96 + t];
    // PHILOX-mode sampling tables + the 1000 possible target SoCs already mapped through each station's curve
    std::vector<float> icdf(4097);
    std::vector<uint32_t> late_thr(16);
    {
        FILE *f = fopen((dir + "/soc_d_icdf_4097.f32").c_str(), "rb");
        if (!f || fread(icdf.data(), 4, 4097, f) != 4097) {
            if (f) fclose(f);
            return bail(fail(CHUB_ERR_DATA, "cannot read soc_d_icdf_4097.f32"));
        }
        fclose(f);
        f = fopen((dir + "/late_thr_16.u32").c_str(), "rb");
        if (!f || fread(late_thr.data(), 4, 16, f) != 16) {
            if (f) fclose(f);
            return bail(fail(CHUB_ERR_DATA, "cannot read late_thr_16.u32"));
        }
        fclose(f);
    }
    std::vector<float> nicdf(4097), ntail(4097);
    {
        FILE *f = fopen((dir + "/normal_icdf_4097.f32").c_str(), "rb");
        if (!f || fread(nicdf.data(), 4, 4097, f) != 4097) {
            if (f) fclose(f);
            return bail(fail(CHUB_ERR_DATA, "cannot read normal_icdf_4097.f32"));
        }
        fclose(f);
        f = fopen((dir + "/normal_tail_4097.f32").c_str(), "rb");
        if (!f || fread(ntail.data(), 4, 4097, f) != 4097) {
            if (f) fclose(f);
            return bail(fail(CHUB_ERR_DATA, "cannot read normal_tail_4097.f32"));
        }
        fclose(f);
    }
    std::vector<double> sin96(96);
    for (int t = 0; t < 96; t++) sin96[t] = sin((2 * M_PI / 96) * (double) t);  // np.sin(k * time), MGR:319-320
    std::vector<float> ttab[2], cls[2], cls_soc0[2];
    const size_t n_classes = (size_t) kSocLevels;
    for (int s = 0; s < 2; s++) {
        const bool fast = hp.type[s] == CHUB_FAST, cpw = hp.constant_charging != 0;
        ttab[s].resize(kLevels);
        float tt_max = 0.0f;
        for (int k = 0; k < kLevels; k++) {
            float tr = (float) k / 999.0f;
            float target = tr * (100.0f - 80.0f) + 80.0f;  // uniform_rand(80, 100), CHS.hpp:35-44
            ttab[s][k] = fast ? fast_soc_to_time(target, cpw) : slow_soc_to_time(target, cpw);
            tt_max = ttab[s][k] > tt_max ? ttab[s][k] : tt_max;
        }
        if (rng_mode != CHUB_RNG_PHILOX) continue;
        // PHILOX: the arrival SoC takes one of kSocLevels = 2048 equiprobable classes: class l sits at probability (l + 0.5) /
        // 2048 of the tabulated inverse CDF of clip(N(7,3),1,10), i.e. exactly at its node 2 l + 1 (mk_soc,
        // CHS.hpp:804-814); its row holds where the car is on its curve after every number of car_steps it can take
        cls[s].assign((n_classes + 1) * (size_t) kClsRow * 2, 0.0f);  // + one row of padding: entry n + 1 is read with entry n
        cls_soc0[s].assign(n_classes, 0.0f);
        float ts_min = 1e30f;
        for (int l = 0; l < kSocLevels; l++) {
            float d = icdf[2 * l + 1];
            if ((double) d < 1.0) d = 1.0f;
            else if ((double) d > 10.0) d = 10.0f;
            const float soc = (float) (75.0 - 5.0 * (double) d);
            cls_soc0[s][l] = soc;
            float *row = &cls[s][(size_t) l * kClsRow * 2];
            build_class_row(fast, cpw, hp.cc, soc, row);
            ts_min = row[1] < ts_min ? row[1] : ts_min;
        }
        // stay_time = ceil(soc_to_time(target) - soc_to_time(soc)) + late (late <= 15): the state word has 5 bits for what is left
        // of it and for the car_steps taken (a car takes at most stay_time - 1), a class row kClsRow = 32 entries
        const int max_stay = (int) ceilf(tt_max - ts_min) + 15;
        if (max_stay > 31) return bail(fail(CHUB_ERR_UNSUPPORTED, "charge curves yield stays longer than 31 slots (the 5-bit fields of the slot state)"));
    }
    e->tape_classes = 0;
    e->tape_pk = nullptr;
    e->tape_car = nullptr;
    e->cur_bits = nullptr;
    e->cur_tail = nullptr;
    e->capturing = false;
    e->graph_base = 0;
    // packed slot kernel (k_slot_packed): the workgroup's virtual lanes laid over whole envs end to end
    {
        const int St = hp.S[0] + hp.S[1];
        // the workgroup tile: the small one while state and action rows live in the caches, the large one once they stream from HBM
        // (a hub too large for the small tile's 512 virtual lanes -- stations of several hundred piles -- still fits the large one's 2048)
        const bool big_tile = opt.tile == 2 || (opt.tile == 0 && (n_envs * (int64_t) St >= kBigTileSlots || St > kPackedBlock * kSlotsPerLane));
        hp.pblock = big_tile ? kBigBlock : kPackedBlock;
        hp.pslots = big_tile ? kBigSlotsPerLane : kSlotsPerLane;
        // ... and the work order: XCD-aware while the streams are cache-resident (measured: 4-6 % of the step; HBM-resident sizes lose 1 %)
        hp.xcd = (opt.work_order == 0 && (!big_tile || CHUB_XCD_ANY_TILE) && n_envs * (int64_t) St <= kXcdOrderSlots) ? 1 : 0;
        const int pb = hp.pblock * hp.pslots;
        hp.epb = pb / St > 0 ? pb / St : 1;
        if (hp.epb > pb / 4) hp.epb = pb / 4;  // the workgroup's per-unit LDS areas hold 2 * pb / 4 units: hubs of 1-3 piles leave lanes idle
        bool magic_ok = true;  // the kernel divides lane numbers by S0 + S1 with a 20-bit reciprocal
        for (int l = 0; l < pb && magic_ok; l++)
            if ((((uint32_t) l * ((1u << 20) / (uint32_t) St + 1u)) >> 20) != (uint32_t) (l / St)) magic_ok = false;
        hp.packed = (rng_mode == CHUB_RNG_PHILOX && St >= 1 && St <= pb && magic_ok &&
                     (uint64_t) n_envs * (uint64_t) (St + 2) * 16u < ((uint64_t) 1 << 32) &&  // 32-bit byte offsets
                     opt.slot_kernel != 1) ? 1 : 0;
    }
    {   // the whole step as one launch: where the two kernels are launch- and latency-bound and every workgroup finds room at once
        const int64_t nb = (n_envs + hp.epb - 1) / hp.epb;
        const bool can = hp.packed && hp.S[0] <= 64 && hp.S[1] <= 64 && hp.pblock == kPackedBlock;
        e->fused = can && (opt.fused_step == 2 || (opt.fused_step == 0 && nb <= (hp.epb <= 64 ? kFusedMaxBlocksTailWave : kFusedMaxBlocks)));
        e->span_size_ok = nb <= kSpanMaxBlocks || opt.fused_step == 2;
        if (opt.fused_step == 2 && !can)
            return bail(fail(CHUB_ERR_UNSUPPORTED, "fused_step = 2: the single-launch step covers PHILOX handles on the packed slot kernel with "
                                                   "stations of at most 64 piles"));
        // chub_run_steps's spans: the tails on a wave of their own, a step behind the slot waves (k_steps_piped) -- its 64 lanes are the workgroup's envs
        const bool can_pipe = e->fused && hp.epb <= 64;
        e->span_piped = can_pipe && (opt.span_tails == 2 || (opt.span_tails == 0 && nb <= kPipedMaxBlocks));
        if (opt.span_tails == 2 && !can_pipe)
            return bail(fail(CHUB_ERR_UNSUPPORTED, "span_tails = 2: the tail wave of a span covers handles on the one-launch step (fused_step) with at most 64 envs "
                                                   "per workgroup (hubs of 8 piles and more)"));
    }
    {   // the reference-exact mode at a handful of envs (the drop-in class: one): both station passes and the tail in one launch
        const int64_t fit = std::min<int64_t>(64, std::min<int64_t>(kCompatSmallWaves0 * (64 / hp.U[0]), kCompatSmallWaves1 * (64 / hp.U[1])));
        e->no_walk_ahead = opt.walk_ahead == 1;
        e->span_steps = opt.span_steps;
        e->compat_small = rng_mode == CHUB_RNG_COMPAT && opt.fused_step != 1 && hp.S[0] <= 64 && hp.S[1] <= 64 && n_envs <= fit;
        // ... and everything else as the split step (stream walks, one env per lane -> slots of both stations in one launch) unless
        // slot_kernel = 1 asks for one kernel per station with the unit's first lane walking (the parity cross-check).  Measured, us per
        // step, split vs per station: 47.1 vs 51.1 at 1024 envs, 49.7 vs 50.9 at 4096, 54 vs 70 at 8192, 100 vs 279 at 65 536 ([20, 25] hub)
        hp.compat_split = (rng_mode == CHUB_RNG_COMPAT && hp.S[0] <= 64 && hp.S[1] <= 64 && opt.slot_kernel != 1) ? 1 : 0;
    }
    build_hy_table(hp, e->hy_table);
    std::vector<double> hy_v(e->hy_table, e->hy_table + 102);

    if ((rc = dev_upload(e, &e->tb.cnt[0], cnt[0]))) return bail(rc);
    if ((rc = dev_upload(e, &e->tb.cnt[1], cnt[1]))) return bail(rc);
    if ((rc = dev_upload(e, &e->tb.cnt_hv, cnt_hv))) return bail(rc);
    if ((rc = dev_upload(e, &e->tb.thr_renege, thr_renege))) return bail(rc);
    if ((rc = dev_upload(e, &e->tb.thr_balk, thr_balk))) return bail(rc);
    if ((rc = dev_upload(e, &e->tb.inv_balk, inv_balk))) return bail(rc);
    if ((rc = dev_upload(e, &e->tb.price, price_v))) return bail(rc);
    if ((rc = dev_upload(e, &e->tb.pvT, pvT))) return bail(rc);
    if ((rc = dev_upload(e, &e->tb.wdT, wdT))) return bail(rc);
    if ((rc = dev_upload(e, &e->tb.hy_table, hy_v))) return bail(rc);
    if ((rc = dev_upload(e, &e->tb.soc_d_icdf, icdf))) return bail(rc);
    if ((rc = dev_upload(e, &e->tb.late_thr, late_thr))) return bail(rc);
    memcpy(e->h_late8, late_thr.data(), sizeof e->h_late8);
    if ((rc = dev_upload(e, &e->tb.normal_icdf, nicdf))) return bail(rc);
    if ((rc = dev_upload(e, &e->tb.normal_tail, ntail))) return bail(rc);
    if ((rc = dev_upload(e, &e->tb.sin96, sin96))) return bail(rc);
    memcpy(e->h_sin96, sin96.data(), sizeof e->h_sin96);
    if ((rc = dev_upload(e, &e->tb.ttab[0], ttab[0]))) return bail(rc);
    if ((rc = dev_upload(e, &e->tb.ttab[1], ttab[1]))) return bail(rc);
    {
        std::vector<float> t2(2048, 0.0f);
        for (int s = 0; s < 2; s++) memcpy(&t2[(size_t) s * 1024], ttab[s].data(), kLevels * sizeof(float));
        if ((rc = dev_upload(e, &e->tb.ttab2, t2))) return bail(rc);
    }
    e->tb.cls[0] = e->tb.cls[1] = e->tb.cls_soc0[0] = e->tb.cls_soc0[1] = nullptr;
    if (rng_mode == CHUB_RNG_PHILOX) {
        // both stations' class tables in one buffer (the packed kernel addresses station 1's as station 0's + a byte distance)
        std::vector<float> both(cls[0]);
        both.insert(both.end(), cls[1].begin(), cls[1].end());
        if ((rc = dev_upload(e, &e->tb.cls[0], both))) return bail(rc);
        e->tb.cls[1] = e->tb.cls[0] + cls[0].size();
        for (int s = 0; s < 2; s++) {
            if ((rc = dev_upload(e, &e->tb.cls_soc0[s], cls_soc0[s]))) return bail(rc);
            e->h_cls[s] = cls[s];
            e->h_soc0[s] = cls_soc0[s];
        }
    }
    for (int s = 0; s < 2; s++) e->h_ttab[s] = ttab[s];  // (both modes: introspection reads a car's target time from its level)

    // ---- state in HBM
    const size_t N = (size_t) n_envs, NS = N * (size_t) (hp.S[0] + hp.S[1]);
#define ALLOC(ptr, count)                                        \
    if ((rc = dev_alloc(e, &(ptr), (count)))) return bail(rc)
    e->sl.stay8 = nullptr;
    e->sl.var[0] = e->sl.var[1] = nullptr;
    e->st.empt = nullptr;
    e->st.fa[0] = e->st.fa[1] = nullptr;
    e->st.empt2[0] = e->st.empt2[1] = e->st.shrt[0] = e->st.shrt[1] = nullptr;
    if (rng_mode == CHUB_RNG_PHILOX) {
        ALLOC(e->sl.hot, NS);  // 4-byte slot state
        ALLOC(e->sl.stay8, NS);
    } else {
        ALLOC(e->sl.hot, 4 * NS);  // the 16-byte hot record: power, arrival SoC, t_soc, word
        ALLOC(e->sl.var[0], 8 * NS); ALLOC(e->sl.var[1], 8 * NS);  // the split step's new cars as the walk made them: 32 bytes per admission rank
        ALLOC(e->st.empt, 2 * N); ALLOC(e->st.fa[0], 2 * N); ALLOC(e->st.fa[1], 2 * N);
        ALLOC(e->st.empt2[0], 2 * N); ALLOC(e->st.empt2[1], 2 * N); ALLOC(e->st.shrt[0], 2 * N); ALLOC(e->st.shrt[1], 2 * N);
    }
    ALLOC(e->st.rec, 8 * N); ALLOC(e->st.pk[0], 2 * N); ALLOC(e->st.pk[1], 2 * N); ALLOC(e->st.tail_act, 2 * N);
    ALLOC(e->ev.cap, N); ALLOC(e->ev.store_soc, N); ALLOC(e->ev.ou, 3 * N); ALLOC(e->ev.price_noise, N);
    ALLOC(e->ev.pv_day, N); ALLOC(e->ev.wd_day, N); ALLOC(e->ev.q_len, N); ALLOC(e->ev.hv_line, N);
    ALLOC(e->ev.q_fold, 2 * N); ALLOC(e->ev.q_fold_cnt, N); ALLOC(e->ev.q_time, N * (size_t) qcap); ALLOC(e->ev.q_mass, N * (size_t) qcap);
    e->ev.hy_env = nullptr;
    ALLOC(e->ev.drw[0], 4 * N); ALLOC(e->ev.drw[1], 4 * N); ALLOC(e->ev.drw_cnt[0], N); ALLOC(e->ev.drw_cnt[1], N);
    e->ev.obs64 = nullptr; e->ev.reward64 = nullptr; e->ev.telem = nullptr;
    for (int p = 0; p < 3; p++) e->cr.g3[p] = e->cr.minstd3[p] = nullptr;
    e->rng_cur = 0;
    e->ev.hv_pre[0] = e->ev.hv_pre[1] = nullptr;
    if (rng_mode == CHUB_RNG_COMPAT) {
        ALLOC(e->ev.hy_env, N * 102);
        for (int p = 0; p < 3; p++) {  // the streams' three buffers in rotation: committed (rng_cur), the walk's shadow, the far walk's (CompatRng)
            ALLOC(e->cr.g3[p], N * 32); ALLOC(e->cr.minstd3[p], N);
        }
        ALLOC(e->ev.hv_pre[0], N * (size_t) hp.hv_w); ALLOC(e->ev.hv_pre[1], N * (size_t) hp.hv_w);
        std::vector<double> rep(N * 102);  // until chub_compat_replay_constructor: every env the zero-demand table
        for (size_t i = 0; i < N; i++) memcpy(&rep[i * 102], e->hy_table, sizeof e->hy_table);
        HIP_TRY(hipMemcpy(e->ev.hy_env, rep.data(), rep.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    ALLOC(e->d_env_clk, 2 * N); ALLOC(e->d_mask, 2 * N);
    ALLOC(e->d_tick_base, 64);
    e->hp.tick_base = e->d_tick_base;
    ALLOC(e->d_ctx, 1);
    e->ctx_dirty = true;
    ALLOC(e->d_actions, N * (size_t) hp.act_dim); ALLOC(e->d_obs, N * (size_t) hp.obs_dim); ALLOC(e->d_reward, N);
    ALLOC(e->d_done, N); ALLOC(e->d_exo_z, N * 3); ALLOC(e->d_exo_days, N * 2);
#undef ALLOC
    {   // tank starts at init_soc (HyStore.__init__, HYD:99-100)
        std::vector<double> cap(N, hp.init_soc * hp.cap_mass), soc(N, hp.init_soc);
        HIP_TRY(hipMemcpy(e->ev.cap, cap.data(), N * sizeof(double), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(e->ev.store_soc, soc.data(), N * sizeof(double), hipMemcpyHostToDevice));
    }
    if (rng_mode == CHUB_RNG_COMPAT) {
        // default seeds: env i gets srand(seed + 2*id + 1), e.seed(seed + 2*id + 2)
        std::vector<uint32_t> seeds(2 * N);
        for (size_t i = 0; i < N; i++) {
            uint32_t id = (uint32_t) (env_id0 + (int64_t) i);
            seeds[2 * i] = (uint32_t) seed + 2u * id + 1u;
            seeds[2 * i + 1] = (uint32_t) seed + 2u * id + 2u;
        }
        *out = e;
        rc = chub_set_rng_compat_seeds(e, seeds.data());
        if (rc) {
            *out = nullptr;
            return bail(rc);
        }
        // the COMPAT hot record keeps a car's target as its LEVEL and reads soc_to_time(target) from Tables::ttab: once per handle the device
        // evaluates all 2 x 1000 entries itself (the functions the stream walk's add_car uses) and the table must hold exactly those bits
        uint32_t *d_bad = nullptr, bad = 0u;
        if ((rc = sync_ctx(e, nullptr))) {
            *out = nullptr;
            return bail(rc);
        }
        if (hipMalloc((void **) &d_bad, sizeof(uint32_t)) != hipSuccess || hipMemset(d_bad, 0, sizeof(uint32_t)) != hipSuccess) {
            if (d_bad) (void) hipFree(d_bad);
            *out = nullptr;
            return bail(fail(CHUB_ERR_HIP, "hipMalloc failed"));
        }
        launch_check_ttab(e->d_ctx, d_bad, nullptr);
        const hipError_t he = hipMemcpy(&bad, d_bad, sizeof(uint32_t), hipMemcpyDeviceToHost);
        (void) hipFree(d_bad);
        if (he != hipSuccess || bad != 0u) {
            *out = nullptr;
            return bail(fail(he != hipSuccess ? CHUB_ERR_HIP : CHUB_ERR_UNSUPPORTED,
                             he != hipSuccess ? "the target-time table could not be checked on the device"
                                              : "the device's soc_to_time differs from the host-built table of target times"));
        }
    }
    *out = e;
    return CHUB_OK;
}

int chub_destroy(chub_env *e) {
    if (!e) return CHUB_OK;
    if (!e->allocs.empty() || !e->prof_events.empty()) {  // a handle that never reached the device owns nothing there
        (void) hipSetDevice(e->device);
        (void) hipDeviceSynchronize();
        for (void *p : e->allocs) (void) hipFree(p);
        for (hipEvent_t ev : e->prof_events) (void) hipEventDestroy(ev);
        if (e->h_actions) (void) hipHostFree(e->h_actions);
        if (e->h_mask) (void) hipHostFree(e->h_mask);
        for (hipEvent_t ev : e->mask_done)
            if (ev) (void) hipEventDestroy(ev);
        if (e->h_packed) (void) hipHostFree(e->h_packed);
        if (e->h_telem) (void) hipHostFree(e->h_telem);
        if (e->d_telem) (void) hipFree(e->d_telem);
        if (e->h_bits) (void) hipHostFree(e->h_bits);  // h_tail / d_tail are the ends of the same blocks
        if (e->d_bits) (void) hipFree(e->d_bits);
        if (e->d_packed) (void) hipFree(e->d_packed);
        if (e->host_stream) (void) hipStreamDestroy(e->host_stream);
        if (e->upload_stream) (void) hipStreamDestroy(e->upload_stream);
    }
    (void) hipGetLastError();  // HIP's last-error slot is per thread and sticky: do not leave ours for the next handle's checks
    delete e;
    return CHUB_OK;
}

int chub_obs_dim(const chub_env *e) { return e ? e->hp.obs_dim : CHUB_ERR_ARG; }
int chub_act_dim(const chub_env *e) { return e ? e->hp.act_dim : CHUB_ERR_ARG; }
int64_t chub_num_envs(const chub_env *e) { return e ? e->hp.n_envs : CHUB_ERR_ARG; }
int chub_clock(const chub_env *e) {  // lock-step: the clock; per-env clocks: env 0's (chub_env_clocks has them all)
    if (!e) return CHUB_ERR_ARG;
    if (!e->per_env) return e->t;
    uint16_t c = 0;
    if (hipSetDevice(e->device) != hipSuccess || hipDeviceSynchronize() != hipSuccess ||
        hipMemcpy(&c, e->d_env_clk + (size_t) ((e->tick + 1u - e->graph_base) & 1u) * (size_t) e->hp.n_envs, sizeof c, hipMemcpyDeviceToHost) != hipSuccess)
        return CHUB_ERR_HIP;
    return (int) (c & 127u);
}
int chub_uses_packed_kernel(const chub_env *e) { return e ? e->hp.packed : CHUB_ERR_ARG; }
int chub_uses_xcd_order(const chub_env *e) { return e ? ((e->hp.packed && e->hp.xcd && !e->fused) ? 1 : 0) : CHUB_ERR_ARG; }  // (the single-launch step has one order only)
int chub_uses_fused_step(const chub_env *e) { return e ? ((e->fused || e->compat_small) ? 1 : 0) : CHUB_ERR_ARG; }

int chub_sync(chub_env *e) {
    if (!e) return fail(CHUB_ERR_ARG, "null handle");
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipDeviceSynchronize());
    return CHUB_OK;
}

// ---- clocks -----------------------------------------------------------------------------------------------------------------
// Every reference env owns its clock (MGR:137-140, 271-273, 299).  Lock-step (the usual case: everybody reset and stepped
// together) is ONE clock on the host, passed to the kernels as an argument.  The first call that names a subset of the envs
// (chub_reset_envs / chub_step_envs) makes the clock per-env device state: [2][N] u16, read at [tick & 1] and written at
// [(tick + 1) & 1] by the tail kernel for every env (served envs: one step on, or back to 0; the others: unchanged), so
// any call is still ONE launch with one Philox tick, however many different clocks the envs show.  A reset of everybody
// returns to the host clock.

// What one call serves: 0 nobody, 1 a subset (mask uploaded, per-env clocks on), 2 every env.
static int serve_mask(chub_env *e, const uint8_t *mask, hipStream_t s, int &served) {
    const size_t N = (size_t) e->hp.n_envs;
    size_t n_masked = N;
    e->mask_lo = 0;
    e->mask_hi = (int64_t) N - 1;
    if (mask) {
        n_masked = 0;
        int64_t lo = -1, hi = -1;
        for (size_t i = 0; i < N; i++)
            if (mask[i]) {
                n_masked++;
                if (lo < 0) lo = (int64_t) i;
                hi = (int64_t) i;
            }
        if (n_masked) {
            e->mask_lo = lo;
            e->mask_hi = hi;
        }
    }
    served = n_masked == 0 ? 0 : (n_masked == N ? 2 : 1);
    if (served != 1) return CHUB_OK;
    if (e->tape_pk) return fail(CHUB_ERR_ARG, "tape mode runs in lock-step");
    if (e->capturing) {
        // a replay must not read the caller's (or the handle's staging) memory: the mask of a captured call lives in a device
        // buffer of its own, filled now and freed with the graph.  The clocks are device state, so nothing of them is baked in.
        if (!e->per_env)
            return fail(CHUB_ERR_ARG, "a capture that names subsets of the envs must start on per-env clocks: make the first call "
                                      "on a subset before chub_graph_begin (it fills the per-env clocks from the handle's clock)");
        uint8_t *d = nullptr;
        HIP_TRY(hipMalloc((void **) &d, N));
        e->cap_masks.push_back(d);
        // filled now, outside the capture: on a stream of its own (the null stream may not be touched while another stream captures)
        if (!e->upload_stream) HIP_TRY(hipStreamCreateWithFlags(&e->upload_stream, hipStreamNonBlocking));
        HIP_TRY(hipMemcpyAsync(d, mask, N, hipMemcpyHostToDevice, e->upload_stream));
        HIP_TRY(hipStreamSynchronize(e->upload_stream));
        e->cur_mask = d;
        return CHUB_OK;
    }
    if (!e->per_env) {  // every env starts from the lock-step clock, in the buffer the next launch reads
        const uint16_t c = (uint16_t) ((uint32_t) e->t | (((uint32_t) e->price_count & 3u) << 8));
        launch_fill_clocks(e->d_env_clk + (size_t) ((e->tick + 1u - e->graph_base) & 1u) * N, (int64_t) N, c, s);
        e->per_env = true;
        if (e->h_tick.size() != N) e->h_tick.assign(N, 0u);
    }
    // the mask of this call: the caller's array is only borrowed for the duration of the call, so it goes through one of two
    // pinned staging buffers; a buffer is reused once the launches that read its device copy are done (two masked calls ago)
    if (!e->h_mask) {
        HIP_TRY(hipHostMalloc((void **) &e->h_mask, 2 * N, hipHostMallocDefault));
        for (hipEvent_t &ev : e->mask_done) HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    }
    const uint32_t b = e->mask_seq & 1u;
    if (e->mask_seq >= 2) HIP_TRY(hipEventSynchronize(e->mask_done[b]));
    memcpy(e->h_mask + (size_t) b * N, mask, N);
    HIP_TRY(hipMemcpyAsync(e->d_mask + (size_t) b * N, e->h_mask + (size_t) b * N, N, hipMemcpyHostToDevice, s));
    e->cur_mask = e->d_mask + (size_t) b * N;
    return CHUB_OK;
}

static int note_served(chub_env *e, const uint8_t *mask, int served, hipStream_t s) {
    const size_t N = (size_t) e->hp.n_envs;
    if (e->capturing) {  // nothing ran: only note which launch of the capture this was (applied by every replay)
        const uint32_t rel = e->tick - e->graph_tick0;
        if (served == 2) e->cap_full_rel = rel;
        else {
            if (e->cap_rel.size() != N) e->cap_rel.assign(N, 0u);
            for (size_t i = 0; i < N; i++)
                if (mask[i]) e->cap_rel[i] = rel;
        }
        return CHUB_OK;
    }
    if (served == 2) {
        e->full_tick = e->tick;
        return CHUB_OK;
    }
    HIP_TRY(hipEventRecord(e->mask_done[e->mask_seq & 1u], s));
    e->mask_seq += 1;
    for (size_t i = 0; i < N; i++)
        if (mask[i]) e->h_tick[i] = e->tick;
    return CHUB_OK;
}

// ONE launched reset: of every env (served = 2) or of the envs of the uploaded mask (served = 1)
// COMPAT: the handle's lock-step steps of every env run the slot pass beside the NEXT step's stream walks (k_slot_walk2)
static bool walks_two_ahead(const chub_env *e) {
    return e->hp.rng_mode == CHUB_RNG_COMPAT && !e->compat_small && !e->no_walk_ahead && slot_walk2_covers(e->hp);
}

static int run_reset(chub_env *e, int served, const int32_t *d_exo_days, const double *d_exo_z, float *d_obs, hipStream_t s) {
    e->tick += 1;
    StepArgs sa;
    memset(&sa, 0, sizeof sa);
    sa.t = 0;
    sa.tick = e->tick - e->graph_base;
    sa.draw_price = (e->price_count % 4 == 0) ? 1 : 0;
    sa.station_filter = -1;
    sa.price_last = e->price[95];  // AGG:171: price = [] + mean_for_MAD; price[-1]
    sa.exo_days = d_exo_days;
    sa.exo_z = d_exo_z;
    sa.obs = d_obs;
    sa.obs_stride = e->hp.obs_dim;
    sa.car_tape = e->tape_car;  // chub_reset_tape: the unit's occupancy draws are in pk already, the cars' variates come from the tape
    sa.tail_tape = e->tape_tail ? 1 : 0;  // chub_reset_tape_env: the tail's days and normals from the caller as well
    // COMPAT split reset: the walk's draws are committed by the slot pass (k_compat_small walks the streams in place: nothing to commit)
    sa.commit_rng = (e->hp.compat_split != 0 && !(e->compat_small && !e->per_env)) ? 1 : 0;
    sa.walk_short = walks_two_ahead(e) ? 1 : 0;  // (the reset's walk leaves its short stays for a walk two steps ahead, as every walk of such a handle)
    sa.rng_cur = e->rng_cur;
    e->walked_tick = 0;                   // (a walk that ran ahead for a step that now does not come: its shadow is simply overwritten)
    sa.env_lo = 0;
    sa.env_hi = (int32_t) (e->hp.n_envs - 1);
    if (e->per_env) {
        sa.env_clk = e->d_env_clk;
        sa.env_mask = served == 1 ? e->cur_mask : nullptr;
        if (served == 1) {
            sa.env_lo = (int32_t) e->mask_lo;
            sa.env_hi = (int32_t) e->mask_hi;
        }
    }
    int rc_ = sync_ctx(e, s);
    if (rc_) return rc_;
    if (e->compat_small && !e->per_env) {
        launch_compat_small(true, e->hp, e->d_ctx, sa, s, packed_ptrs(e));
        e->empt_valid = true;  // (k_compat_small is the split step in one launch: its slot waves leave the counts)
    } else {
        launch_slot(true, e->hp, e->d_ctx, sa, s, packed_ptrs(e), nullptr, nullptr);
        if (sa.commit_rng) sa.rng_cur = e->rng_cur = (e->rng_cur + 1) % 3;  // the commit: the walk's shadow is the streams' buffer from here on
        launch_env(true, e->hp, e->d_ctx, sa, s, nullptr, nullptr, packed_ptrs(e));
        // (a split reset leaves the counts of the units it served; those of the others are as good as they were)
        e->empt_valid = e->hp.compat_split != 0 && (served == 2 || e->empt_valid) && !e->capturing;
        if (e->hp.compat_split != 0 && served == 2 && !e->capturing) e->e2_tick = e->tick;  // (... and every unit's empt2, at this tick's parity)
    }
    HIP_TRY(hipGetLastError());
    e->predrawn = served == 2;  // the launch's level blocks left the next step's draws of every env it served
    if (served == 2) {  // everybody starts a new day: one clock again (the launch itself still ran on the envs' own clocks)
        e->per_env = false;
        e->t = 0;
        e->price_count = 0;  // MGR:313 (after make_state)
    }
    return CHUB_OK;
}

static int reset_masked(chub_env *e, const uint8_t *mask, const int32_t *d_exo_days, const double *d_exo_z, float *d_obs, void *stream) {
    if (!e || !d_obs) return fail(CHUB_ERR_ARG, "null argument");
    if (e->hp.rng_mode == CHUB_RNG_COMPAT && (!d_exo_days || !d_exo_z))
        return fail(CHUB_ERR_ARG, "COMPAT mode needs exo_days and exo_z");
    if (e->tape_only && !e->tape_car)
        return fail(CHUB_ERR_ARG, "a handle with registered tape classes resets through chub_reset_tape / chub_reset_tape_env only");
    HIP_TRY(hipSetDevice(e->device));
    (void) hipGetLastError();  // a stale error of an earlier, unrelated call must not be reported as this step's
    hipStream_t s = (hipStream_t) stream;
    int served = 0;
    const bool was_per_env = e->per_env;
    int rc = serve_mask(e, mask, s, served);
    if (rc || served == 0) return rc;
    rc = run_reset(e, served, d_exo_days, d_exo_z, d_obs, s);
    if (rc) {
        e->per_env = was_per_env;  // nothing was launched: the handle stays on the clock(s) it was on
        return rc;
    }
    return note_served(e, mask, served, s);
}

int chub_reset_device(chub_env *e, const int32_t *d_exo_days, const double *d_exo_z, float *d_obs, void *stream) {
    return reset_masked(e, nullptr, d_exo_days, d_exo_z, d_obs, stream);
}

int chub_reset_envs_device(chub_env *e, const uint8_t *mask, const int32_t *d_exo_days, const double *d_exo_z, float *d_obs, void *stream) {
    if (!mask) return fail(CHUB_ERR_ARG, "null argument");
    return reset_masked(e, mask, d_exo_days, d_exo_z, d_obs, stream);
}

static int step_common(chub_env *e, const float *d_actions, const double *d_exo_z, float *d_obs, int obs_stride,
                       float *d_reward, int reward_stride, uint8_t *d_done, float *d_done_f32, void *stream,
                       int load_mode = 0);

int chub_step_device(chub_env *e, const float *d_actions, const double *d_exo_z, float *d_obs, float *d_reward,
                     uint8_t *d_done, void *stream) {
    if (!e || !d_actions || !d_obs || !d_reward || !d_done) return fail(CHUB_ERR_ARG, "null argument");
    return step_common(e, d_actions, d_exo_z, d_obs, e->hp.obs_dim, d_reward, 1, d_done, nullptr, stream);
}

int chub_step_device_packed(chub_env *e, const float *d_actions, const double *d_exo_z, float *d_packed, void *stream) {
    if (!e || !d_actions || !d_packed) return fail(CHUB_ERR_ARG, "null argument");
    const int D = e->hp.obs_dim;
    return step_common(e, d_actions, d_exo_z, d_packed, D + 2, d_packed + D, D + 2, nullptr, d_packed + D + 1, stream);
}

int chub_step_gather(chub_env *e, chub_comm *comm, const float *d_actions, float *d_packed, float *d_gathered, void *stream) {
    if (!e || !comm || !d_actions) return fail(CHUB_ERR_ARG, "null argument");
    // (overlapped gathers, chub_comm_set_overlap: the gather that last sent this block must have left before the kernels overwrite it;
    // a capture remembers the communicator so that chub_graph_end can join its stream)
    // the ROOT's kernels write its packed block straight into the gathered buffer (its first n_envs rows): no copy of the root's own block, neither
    // by a send to itself nor otherwise; d_packed is not touched there (may be null).  Every other rank steps into d_packed and sends it.
    const bool root = chub_comm_rank(comm) == 0;
    if (root && !d_gathered) return fail(CHUB_ERR_ARG, "rank 0 needs the gathered buffer");
    if (!root && !d_packed) return fail(CHUB_ERR_ARG, "null argument");
    float *out = root ? d_gathered : d_packed;
    int rc = chub_comm_gather_begin(comm, out, stream);
    if (rc) return rc;
    if (e->capturing) e->cap_comm = comm;
    rc = chub_step_device_packed(e, d_actions, nullptr, out, stream);
    if (rc) return rc;
    return chub_comm_gather(comm, out, d_gathered, e->hp.n_envs * (int64_t) (e->hp.obs_dim + 2) * (int64_t) sizeof(float), stream);
}

// chub_run_steps on a PHILOX handle that runs the one-launch step (k_step_fused): a span of lock-step steps goes out as ONE launch
// (k_steps_fused: the workgroup that owns an env's slots, records and tail goes from step to step by itself).  Not while a tape is loaded,
// on per-env clocks, with one bit per pile, under the per-kernel profiler or with chub_options.span_steps = 1.
static bool span_ok(const chub_env *e, int n_batches) {
    return e->fused && e->span_size_ok && e->span_steps != 1 && e->hp.rng_mode == CHUB_RNG_PHILOX && !e->per_env && !e->prof_on && !e->tape_pk && !e->tape_car &&
           !e->tape_tail && !e->tape_only && !e->cur_bits && n_batches <= 8 && e->tick != 0 && !e->hp.telemetry;
}

static int run_span(chub_env *e, const float *const *batches, int n_batches, float *const *packed2, int64_t first, int k, hipStream_t s) {
    HIP_TRY(hipSetDevice(e->device));
    (void) hipGetLastError();
    if (e->t + k > 96) return fail(CHUB_ERR_ARG, "a span of steps ends at the day's end at the latest");
    const int D = e->hp.obs_dim;
    StepArgs sa;
    memset(&sa, 0, sizeof sa);
    sa.t = e->t;
    sa.tick = e->tick + 1u - e->graph_base;
    sa.draw_price = (e->price_count % 4 == 0) ? 1 : 0;
    sa.station_filter = -1;
    sa.price_last = e->price[e->t];
    sa.price_prev = e->price[(e->t + 95) % 96];
    sa.actions = batches[first % n_batches];
    float *out = packed2[first & 1];
    sa.obs = out;
    sa.obs_stride = D + 2;
    sa.reward = out + D;
    sa.reward_stride = D + 2;
    sa.done_f32 = out + D + 1;
    sa.env_lo = 0;
    sa.env_hi = (int32_t) (e->hp.n_envs - 1);
    sa.fresh = (!e->predrawn || (e->capturing && e->tick == e->graph_tick0)) ? 1 : 0;  // (as run_step: a graph's first step makes its own draws)
#if CHUB_TRACE
    sa.stamps_slot = nullptr;
    sa.stamps_env = nullptr;
#endif
    int rc = sync_ctx(e, s);
    if (rc) return rc;
    launch_steps_fused(e->hp, e->d_ctx, sa, s, packed_ptrs(e), k, e->price_count, first, batches, n_batches, packed2, e->span_piped);
    HIP_TRY(hipGetLastError());
    e->tick += (uint32_t) k;
    if (e->capturing) e->cap_full_rel = e->tick - e->graph_tick0;  // (note_served: which launch of the capture served every env last)
    else e->full_tick = e->tick;
    e->predrawn = true;
    e->price_count += k;
    e->t = (e->t + k) % 96;
    return CHUB_OK;
}

// A run of steps issued from C: what a host loop of chub_reset_device / chub_step_device_packed / chub_step_gather calls does, without
// a trip through the host language per step (multi-GPU shards of a few thousand envs are otherwise bound by the host's issue rate)
int chub_run_steps(chub_env *e, chub_comm *comm, const float *const *d_action_batches, int n_batches, float *const *d_packed2,
                   float *const *d_gathered2, float *d_reset_obs, int64_t first_step, int64_t n_steps, void *stream) {
    if (!e || !d_action_batches || n_batches <= 0 || !d_packed2 || !d_packed2[0] || !d_packed2[1] || !d_reset_obs || first_step < 0 ||
        n_steps < 0)
        return fail(CHUB_ERR_ARG, "bad argument");
    for (int64_t i = first_step; i < first_step + n_steps; i++) {
        int rc;
        if (i % 96 == 0 && (rc = chub_reset_device(e, nullptr, nullptr, d_reset_obs, stream))) return rc;
        // a SPAN of steps in ONE launch (k_steps_fused) where the handle runs the one-launch step anyway: up to the day's end or the call's
        if (!comm) {
            int64_t k = first_step + n_steps - i;
            k = k < 96 - i % 96 ? k : 96 - i % 96;
            k = k < 96 - e->t ? k : 96 - e->t;  // (the handle's own clock need not be i % 96: a caller may step on past `done` -- the span ends where the clock wraps)
            if (e->span_steps > 1 && k > e->span_steps) k = e->span_steps;
            if (k >= 2 && span_ok(e, n_batches)) {
                if ((rc = run_span(e, d_action_batches, n_batches, d_packed2, i, (int) k, (hipStream_t) stream))) return rc;
                i += k - 1;
                continue;
            }
        }
        const float *act = d_action_batches[i % n_batches];
        float *packed = d_packed2[i & 1];
        if (comm) rc = chub_step_gather(e, comm, act, packed, d_gathered2 ? d_gathered2[i & 1] : nullptr, stream);
        else rc = chub_step_device_packed(e, act, nullptr, packed, stream);
        if (rc) return rc;
    }
    return CHUB_OK;
}

int chub_step_load_device(chub_env *e, const float *d_actions, const double *d_exo_z, float *d_obs, float *d_reward,
                          uint8_t *d_done, void *stream) {
    if (!e || !d_actions || !d_obs || !d_reward || !d_done) return fail(CHUB_ERR_ARG, "null argument");
    return step_common(e, d_actions, d_exo_z, d_obs, e->hp.obs_dim, d_reward, 1, d_done, nullptr, stream, 1);
}

int chub_step_load(chub_env *e, const float *actions, const double *exo_z, float *obs, float *reward, uint8_t *done) {
    if (!e || !actions || !obs || !reward || !done) return fail(CHUB_ERR_ARG, "null argument");
    HIP_TRY(hipSetDevice(e->device));
    const size_t N = (size_t) e->hp.n_envs;
    HIP_TRY(hipMemcpy(e->d_actions, actions, N * (size_t) e->hp.act_dim * sizeof(float), hipMemcpyHostToDevice));
    if (e->hp.rng_mode == CHUB_RNG_COMPAT) {
        if (!exo_z) return fail(CHUB_ERR_ARG, "COMPAT mode needs exo_z");
        HIP_TRY(hipMemcpy(e->d_exo_z, exo_z, N * 3 * sizeof(double), hipMemcpyHostToDevice));
    }
    int rc = chub_step_load_device(e, e->d_actions, e->d_exo_z, e->d_obs, e->d_reward, e->d_done, nullptr);
    if (rc) return rc;
    HIP_TRY(hipMemcpy(obs, e->d_obs, N * (size_t) e->hp.obs_dim * sizeof(float), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(reward, e->d_reward, N * sizeof(float), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(done, e->d_done, N, hipMemcpyDeviceToHost));
    return CHUB_OK;
}

// ONE launched step: of every env (served = 2) or of the envs of the uploaded mask (served = 1)
static int run_step(chub_env *e, int served, const float *d_actions, const double *d_exo_z, float *d_obs, int obs_stride,
                    float *d_reward, int reward_stride, uint8_t *d_done, float *d_done_f32, hipStream_t s, int load_mode) {
    e->tick += 1;
    StepArgs sa;
    memset(&sa, 0, sizeof sa);
    sa.t = e->t;
    sa.tick = e->tick - e->graph_base;
    sa.draw_price = (e->price_count % 4 == 0) ? 1 : 0;
    sa.station_filter = -1;
    sa.price_last = e->price[e->t];  // AGG:147
    sa.price_prev = e->price[(e->t + 95) % 96];  // what the make_state before this step saw there (the reset: price[95], AGG:171)
    sa.actions = d_actions;
    sa.act_bits = e->cur_bits;
    sa.act_tail = e->cur_tail;
    sa.exo_z = d_exo_z;
    sa.obs = d_obs;
    sa.obs_stride = obs_stride;
    sa.reward = d_reward;
    sa.reward_stride = reward_stride;
    sa.done = d_done;
    sa.done_f32 = d_done_f32;
    sa.load_mode = load_mode;
    sa.pk_tape = e->tape_pk;
    sa.car_tape = e->tape_car;
#if CHUB_TRACE
    sa.stamps_slot = (unsigned long long *) g_stamps_slot;
    sa.stamps_env = (unsigned long long *) g_stamps_env;
#endif
    sa.hv_tape = e->tape_hv;
    sa.hv_w = e->tape_hv_w;
    sa.tail_tape = e->tape_tail ? 1 : 0;
    const bool split_step = e->hp.compat_split != 0 && !(e->compat_small && !load_mode && !e->per_env);
    sa.rng_cur = e->rng_cur;
    if (split_step) {  // COMPAT split step: the slot pass commits the walk's draws, the tail reads the forecourt's from where the walk left them
        sa.commit_rng = 1;
        sa.hv_tape = (const uint32_t *) e->ev.hv_pre[sa.tick & 1u];
        sa.hv_w = e->hp.hv_w;
        sa.walked = (e->walked_tick == e->tick && served == 2 && !e->per_env) ? 1 : 0;
        sa.walk_short = walks_two_ahead(e) ? 1 : 0;
    }
    e->walked_tick = 0;
    sa.env_lo = 0;
    sa.env_hi = (int32_t) (e->hp.n_envs - 1);
    if (e->per_env) {
        sa.env_clk = e->d_env_clk;
        sa.env_mask = served == 1 ? e->cur_mask : nullptr;
        if (served == 1) {
            sa.env_lo = (int32_t) e->mask_lo;
            sa.env_hi = (int32_t) e->mask_hi;
        }
    }
    // the state-independent draws of this step: left by the previous launch's level blocks if that launch served every env
    // (for the tick that is now this launch's), otherwise made by this launch itself (a graph's first step always makes its
    // own: a replay must not depend on what ran before it)
    sa.fresh = (!e->predrawn || (e->capturing && e->tick == e->graph_tick0 + 1u)) ? 1 : 0;
    int rc_ = sync_ctx(e, s);
    if (rc_) return rc_;
    bool prof = e->prof_on && e->prof_used < e->prof_cap;
    if (prof) {
        prof = (e->prof_phase % e->prof_every) == 0;  // sample: the event records are not free
        e->prof_phase++;
    }
    // four events per profiled step: start / stop of the slot kernel, start / stop of the tail kernel
    hipEvent_t *pe = prof ? &e->prof_events[4 * e->prof_used] : nullptr;
    const bool small_step = e->compat_small && !load_mode && !e->per_env;
    sa.empt_fresh = ((small_step || e->hp.compat_split) && (!e->empt_valid || e->capturing)) ? 1 : 0;
    if (small_step) {
        launch_compat_small(false, e->hp, e->d_ctx, sa, s, packed_ptrs(e));
        e->empt_valid = true;
        if (prof) {  // one kernel, no dispatch timestamps: the sample spans nothing
            for (int i = 0; i < 4; i++) HIP_TRY(hipEventRecord(pe[i], s));
        }
    } else if (e->fused && !load_mode && !e->per_env && ((!sa.car_tape && !sa.pk_tape) || (sa.car_tape && sa.pk_tape && sa.tail_tape))) {
        // (tape mode: the one-launch form replays only a complete tape -- station draws, car variates and the tail's variates)
        launch_step_fused(e->hp, e->d_ctx, sa, s, packed_ptrs(e), prof ? pe[0] : nullptr, prof ? pe[1] : nullptr);
        if (prof) {  // one kernel: the whole step is on the first pair of timestamps, the second pair spans nothing
            HIP_TRY(hipEventRecord(pe[2], s));
            HIP_TRY(hipEventRecord(pe[3], s));
        }
    } else if (split_step && served == 2 && !e->per_env && !load_mode && !e->capturing && walks_two_ahead(e)) {
        // lock-step COMPAT steps of every env, stations of 8 to 64 piles: the slot pass of this step beside the stream walks of the NEXT one
        // (k_slot_walk2: the walk two steps ahead of the slots it draws for), then the tails alone -- if the next call is that step, its walk
        // has run; if it is anything else, the walk's shadow is never committed
        if (e->e2_tick != e->tick - 1u) sa.empt_fresh = 1;  // (empt2 is not the previous pass's for every unit: counted in front)
        StepArgs sw = sa;
        sw.t = (e->t + 1) % 96;
        sw.tick = sa.tick + 1u;
        sw.walk_far = 1;  // (reads the buffer behind the committed one -- this step's shadow -- and writes the one behind that)
        launch_slot_walk2(e->hp, e->d_ctx, sa, sw, s, prof ? pe[0] : nullptr, prof ? pe[1] : nullptr);
        sa.rng_cur = e->rng_cur = (e->rng_cur + 1) % 3;  // the commit of this step's draws: its walk's shadow is the streams' buffer now
        launch_env(false, e->hp, e->d_ctx, sa, s, prof ? pe[2] : nullptr, prof ? pe[3] : nullptr, packed_ptrs(e));
        e->walked_tick = e->tick + 1u;
        e->empt_valid = true;
        e->e2_tick = e->tick;
    } else {
        launch_slot(false, e->hp, e->d_ctx, sa, s, packed_ptrs(e), prof ? pe[0] : nullptr, prof ? pe[1] : nullptr);
        if (sa.commit_rng) sa.rng_cur = e->rng_cur = (e->rng_cur + 1) % 3;  // the commit (as above)
        if (split_step && served == 2 && !e->per_env && !e->capturing) e->e2_tick = e->tick;  // (every unit's empt2, whichever pass it was)
        if (split_step && served == 2 && !e->per_env && !e->no_walk_ahead) {
            // lock-step COMPAT steps of every env: the tails of this step and the stream walks of the NEXT one in one launch (k_env_walk) --
            // if the next call is that step, its walk has run; if it is anything else, the walk's shadow is never committed
            StepArgs sw = sa;  // (rng_cur: the buffer this step's slot pass has just made the committed one)
            sw.t = (e->t + 1) % 96;
            sw.tick = sa.tick + 1u;
            launch_env_walk(e->hp, e->d_ctx, sa, sw, s, prof ? pe[2] : nullptr, prof ? pe[3] : nullptr, packed_ptrs(e));
            e->walked_tick = e->tick + 1u;
        } else {
            launch_env(false, e->hp, e->d_ctx, sa, s, prof ? pe[2] : nullptr, prof ? pe[3] : nullptr, packed_ptrs(e));
        }
        e->empt_valid = e->hp.compat_split != 0 && !e->capturing;  // (counted for every unit in front of the walk, or good already; the pass left the served units')
    }
    if (prof) e->prof_used++;
    HIP_TRY(hipGetLastError());
    e->predrawn = served == 2;
    if (!e->per_env) {
        e->price_count += 1;
        e->t = (e->t + 1) % 96;
    }
    return CHUB_OK;
}

static int step_masked(chub_env *e, const uint8_t *mask, const float *d_actions, const double *d_exo_z, float *d_obs, int obs_stride,
                       float *d_reward, int reward_stride, uint8_t *d_done, float *d_done_f32, void *stream, int load_mode) {
    if (e->tick == 0) return fail(CHUB_ERR_ARG, "step() before reset()");
    if (e->tape_only && !e->tape_pk)
        return fail(CHUB_ERR_ARG, "a handle with registered tape classes steps through chub_step_tape / chub_step_tape_env only (its class "
                                  "rows hold the caller's arrival SoCs: cars admitted by this build's own draws would be given them)");
    if (e->hp.rng_mode == CHUB_RNG_COMPAT && !d_exo_z) return fail(CHUB_ERR_ARG, "COMPAT mode needs exo_z");
    HIP_TRY(hipSetDevice(e->device));
    (void) hipGetLastError();  // a stale error of an earlier, unrelated call must not be reported as this step's
    hipStream_t s = (hipStream_t) stream;
    int served = 0;
    const bool was_per_env = e->per_env;
    int rc = serve_mask(e, mask, s, served);
    if (rc || served == 0) return rc;
    rc = run_step(e, served, d_actions, d_exo_z, d_obs, obs_stride, d_reward, reward_stride, d_done, d_done_f32, s, load_mode);
    if (rc) {
        e->per_env = was_per_env;  // nothing was launched: the handle stays on the clock(s) it was on
        return rc;
    }
    return note_served(e, mask, served, s);
}

static int step_common(chub_env *e, const float *d_actions, const double *d_exo_z, float *d_obs, int obs_stride,
                       float *d_reward, int reward_stride, uint8_t *d_done, float *d_done_f32, void *stream,
                       int load_mode) {
    return step_masked(e, nullptr, d_actions, d_exo_z, d_obs, obs_stride, d_reward, reward_stride, d_done, d_done_f32, stream, load_mode);
}

int chub_step_envs_device(chub_env *e, const uint8_t *mask, const float *d_actions, const double *d_exo_z, float *d_obs, float *d_reward,
                          uint8_t *d_done, void *stream) {
    if (!e || !mask || !d_actions || !d_obs || !d_reward || !d_done) return fail(CHUB_ERR_ARG, "null argument");
    return step_masked(e, mask, d_actions, d_exo_z, d_obs, e->hp.obs_dim, d_reward, 1, d_done, nullptr, stream, 0);
}

// the scalar-load step on a subset of the envs (every reference env can take evs_step(float) on its own, CHS.hpp:1169-1186 / 1480-1497)
int chub_step_load_envs_device(chub_env *e, const uint8_t *mask, const float *d_actions, const double *d_exo_z, float *d_obs, float *d_reward,
                               uint8_t *d_done, void *stream) {
    if (!e || !mask || !d_actions || !d_obs || !d_reward || !d_done) return fail(CHUB_ERR_ARG, "null argument");
    return step_masked(e, mask, d_actions, d_exo_z, d_obs, e->hp.obs_dim, d_reward, 1, d_done, nullptr, stream, 1);
}

// host-pointer forms: full-size arrays, only the rows of the named envs are read and written
int chub_reset_envs(chub_env *e, const uint8_t *mask, const int32_t *exo_days, const double *exo_z, float *obs) {
    if (!e || !mask || !obs) return fail(CHUB_ERR_ARG, "null argument");
    HIP_TRY(hipSetDevice(e->device));
    const size_t N = (size_t) e->hp.n_envs, D = (size_t) e->hp.obs_dim;
    if (e->hp.rng_mode == CHUB_RNG_COMPAT) {  // as chub_reset: the values of the envs outside the mask are not looked at
        if (!exo_days || !exo_z) return fail(CHUB_ERR_ARG, "COMPAT mode needs exo_days and exo_z");
        for (size_t i = 0; i < N; i++)
            if (mask[i] && (exo_days[2 * i] < 0 || exo_days[2 * i] >= 100 || exo_days[2 * i + 1] < 0 || exo_days[2 * i + 1] >= 150))
                return fail(CHUB_ERR_ARG, "exo_days out of range");
        HIP_TRY(hipMemcpy(e->d_exo_days, exo_days, N * 2 * sizeof(int32_t), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(e->d_exo_z, exo_z, N * 3 * sizeof(double), hipMemcpyHostToDevice));
    }
    int rc = chub_reset_envs_device(e, mask, e->d_exo_days, e->d_exo_z, e->d_obs, nullptr);
    if (rc) return rc;
    std::vector<float> o(N * D);
    HIP_TRY(hipMemcpy(o.data(), e->d_obs, o.size() * sizeof(float), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < N; i++)
        if (mask[i]) memcpy(obs + i * D, &o[i * D], D * sizeof(float));
    return CHUB_OK;
}

static int step_envs_host(chub_env *e, const uint8_t *mask, const float *actions, const double *exo_z, float *obs, float *reward, uint8_t *done,
                          int load_mode) {
    if (!e || !mask || !actions || !obs || !reward || !done) return fail(CHUB_ERR_ARG, "null argument");
    HIP_TRY(hipSetDevice(e->device));
    const size_t N = (size_t) e->hp.n_envs, A = (size_t) e->hp.act_dim, D = (size_t) e->hp.obs_dim;
    HIP_TRY(hipMemcpy(e->d_actions, actions, N * A * sizeof(float), hipMemcpyHostToDevice));
    if (e->hp.rng_mode == CHUB_RNG_COMPAT) {
        if (!exo_z) return fail(CHUB_ERR_ARG, "COMPAT mode needs exo_z");
        HIP_TRY(hipMemcpy(e->d_exo_z, exo_z, N * 3 * sizeof(double), hipMemcpyHostToDevice));
    }
    int rc = load_mode ? chub_step_load_envs_device(e, mask, e->d_actions, e->d_exo_z, e->d_obs, e->d_reward, e->d_done, nullptr)
                       : chub_step_envs_device(e, mask, e->d_actions, e->d_exo_z, e->d_obs, e->d_reward, e->d_done, nullptr);
    if (rc) return rc;
    std::vector<float> o(N * D), r(N);
    std::vector<uint8_t> d(N);
    HIP_TRY(hipMemcpy(o.data(), e->d_obs, o.size() * sizeof(float), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(r.data(), e->d_reward, N * sizeof(float), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(d.data(), e->d_done, N, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < N; i++)
        if (mask[i]) {
            memcpy(obs + i * D, &o[i * D], D * sizeof(float));
            reward[i] = r[i];
            done[i] = d[i];
        }
    return CHUB_OK;
}

int chub_step_envs(chub_env *e, const uint8_t *mask, const float *actions, const double *exo_z, float *obs, float *reward, uint8_t *done) {
    return step_envs_host(e, mask, actions, exo_z, obs, reward, done, 0);
}

int chub_step_load_envs(chub_env *e, const uint8_t *mask, const float *actions, const double *exo_z, float *obs, float *reward, uint8_t *done) {
    return step_envs_host(e, mask, actions, exo_z, obs, reward, done, 1);
}

// slot of day (and, if asked, the Philox tick of the last launch) of every env
static int fetch_clocks(chub_env *e, std::vector<uint16_t> &c) {
    const size_t N = (size_t) e->hp.n_envs;
    c.assign(N, (uint16_t) ((uint32_t) e->t | (((uint32_t) e->price_count & 3u) << 8)));
    if (!e->per_env) return CHUB_OK;
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipDeviceSynchronize());
    // the buffer the next launch reads
    HIP_TRY(hipMemcpy(c.data(), e->d_env_clk + (size_t) ((e->tick + 1u - e->graph_base) & 1u) * N, N * sizeof(uint16_t), hipMemcpyDeviceToHost));
    return CHUB_OK;
}

int chub_env_clocks(chub_env *e, int32_t *t_out, uint32_t *tick_out) {
    if (!e || !t_out) return fail(CHUB_ERR_ARG, "null argument");
    const size_t N = (size_t) e->hp.n_envs;
    std::vector<uint16_t> c;
    int rc = fetch_clocks(e, c);
    if (rc) return rc;
    for (size_t i = 0; i < N; i++) {
        t_out[i] = (int32_t) (c[i] & 127u);
        if (tick_out) tick_out[i] = (e->h_tick.size() == N && e->h_tick[i] > e->full_tick) ? e->h_tick[i] : e->full_tick;
    }
    return CHUB_OK;
}

int chub_clock_groups(chub_env *e) {  // number of distinct clocks among the envs
    if (!e) return CHUB_ERR_ARG;
    std::vector<uint16_t> c;
    int rc = fetch_clocks(e, c);
    if (rc) return rc;
    std::sort(c.begin(), c.end());
    return (int) (std::unique(c.begin(), c.end()) - c.begin());
}

int chub_reset(chub_env *e, const int32_t *exo_days, const double *exo_z, float *obs) {
    if (!e || !obs) return fail(CHUB_ERR_ARG, "null argument");
    HIP_TRY(hipSetDevice(e->device));
    const size_t N = (size_t) e->hp.n_envs;
    if (e->hp.rng_mode == CHUB_RNG_COMPAT) {
        if (!exo_days || !exo_z) return fail(CHUB_ERR_ARG, "COMPAT mode needs exo_days and exo_z");
        for (size_t i = 0; i < N; i++)
            if (exo_days[2 * i] < 0 || exo_days[2 * i] >= 100 || exo_days[2 * i + 1] < 0 || exo_days[2 * i + 1] >= 150)
                return fail(CHUB_ERR_ARG, "exo_days out of range");
        HIP_TRY(hipMemcpy(e->d_exo_days, exo_days, N * 2 * sizeof(int32_t), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(e->d_exo_z, exo_z, N * 3 * sizeof(double), hipMemcpyHostToDevice));
    }
    int rc = chub_reset_device(e, e->d_exo_days, e->d_exo_z, e->d_obs, nullptr);
    if (rc) return rc;
    HIP_TRY(hipMemcpy(obs, e->d_obs, N * (size_t) e->hp.obs_dim * sizeof(float), hipMemcpyDeviceToHost));
    return CHUB_OK;
}

// The output arrays of a host-pointer step, as the device sees them: pinned host memory (hipHostMalloc: chub_alloc_host, what
// VecChargingHub hands in) is mapped into the device's address space, so the tail kernel can store its rows straight into the
// caller's arrays -- they cross PCIe as posted writes while the kernel runs instead of in a copy phase of their own behind it.
// Any other host memory gets the handle's device staging and a copy back.
static bool device_view(void *host, void **dev) {
    void *d = nullptr;
    if (hipHostGetDevicePointer(&d, host, 0) != hipSuccess || !d) {
        (void) hipGetLastError();
        return false;
    }
    *dev = d;
    return true;
}

// Host-pointer step: the reference-shaped API (numpy in / numpy out), bounded by PCIe: (A + D + 2) * 4 bytes per env and
// step.  A private stream; the actions go up from the handle's pinned buffer when the caller filled that one
// (chub_host_actions: one DMA, no staging copy), otherwise through the runtime's pageable-copy path; outputs come back
// into the caller's arrays.
static int host_path_init(chub_env *e) {
    if (e->host_stream) return CHUB_OK;
    const size_t N = (size_t) e->hp.n_envs, A = (size_t) e->hp.act_dim;
    HIP_TRY(hipStreamCreateWithFlags(&e->host_stream, hipStreamNonBlocking));
    HIP_TRY(hipHostMalloc((void **) &e->h_actions, N * A * sizeof(float), hipHostMallocDefault));
    return CHUB_OK;
}

int chub_host_actions(chub_env *e, float **out) {
    if (!e || !out) return fail(CHUB_ERR_ARG, "null argument");
    HIP_TRY(hipSetDevice(e->device));
    int rc = host_path_init(e);
    if (rc) return rc;
    *out = e->h_actions;
    return CHUB_OK;
}

int chub_step(chub_env *e, const float *actions, const double *exo_z, float *obs, float *reward, uint8_t *done) {
    if (!e || !actions || !obs || !reward || !done) return fail(CHUB_ERR_ARG, "null argument");
    HIP_TRY(hipSetDevice(e->device));
    int rc = host_path_init(e);
    if (rc) return rc;
    const size_t N = (size_t) e->hp.n_envs, A = (size_t) e->hp.act_dim, D = (size_t) e->hp.obs_dim;
    hipStream_t s = e->host_stream;
    // A handful of envs (the drop-in class: one): inputs that sit in pinned host memory are READ BY THE KERNELS where they are -- a
    // few hundred bytes over PCIe -- instead of going through a copy engine first: the step is then one launch set and one
    // stream synchronisation, no copy in either direction.  Larger batches and pageable memory take the DMA as before.
    const bool tiny = N * A * sizeof(float) <= 16384;
    const double *d_z = e->d_exo_z;
    if (e->hp.rng_mode == CHUB_RNG_COMPAT) {
        if (!exo_z) return fail(CHUB_ERR_ARG, "COMPAT mode needs exo_z");
        void *vz = nullptr;
        if (tiny && device_view((void *) exo_z, &vz)) d_z = (const double *) vz;
        else HIP_TRY(hipMemcpyAsync(e->d_exo_z, exo_z, N * 3 * sizeof(double), hipMemcpyHostToDevice, s));
    }
    const float *d_act = e->d_actions;
    void *va = nullptr;
    if (tiny && device_view((void *) actions, &va)) d_act = (const float *) va;
    else  // from the handle's pinned buffer (chub_host_actions) this is one DMA; any other host pointer is staged by the HIP runtime
        HIP_TRY(hipMemcpyAsync(e->d_actions, actions, N * A * sizeof(float), hipMemcpyHostToDevice, s));
    void *v_obs = nullptr, *v_rew = nullptr, *v_done = nullptr;
    if (device_view(obs, &v_obs) && device_view(reward, &v_rew) && device_view(done, &v_done)) {
        // pinned output arrays: the tail kernel stores into them directly (see device_view)
        rc = chub_step_device(e, d_act, d_z, (float *) v_obs, (float *) v_rew, (uint8_t *) v_done, s);
        if (rc) return rc;
    } else {
        rc = chub_step_device(e, d_act, d_z, e->d_obs, e->d_reward, e->d_done, s);
        if (rc) return rc;
        HIP_TRY(hipMemcpyAsync(obs, e->d_obs, N * D * sizeof(float), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipMemcpyAsync(reward, e->d_reward, N * sizeof(float), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipMemcpyAsync(done, e->d_done, N, hipMemcpyDeviceToHost, s));
    }
    HIP_TRY(hipStreamSynchronize(s));
    return CHUB_OK;
}

// ---- packed-action form of the host-pointer step: one bit per pile + the two tail floats (16 bytes per env for hubs of up to
// 64 piles instead of 4 * (S + 2)): what action_to_real (MGR:384-393) keeps of an action row.  The packed slot kernel reads
// the bits themselves (k_slot_packed<.., BITS>: the same step with another action load, bit for bit chub_step's results); for the
// other kernels they are expanded on the device into the action rows those read.
static int bits_path_init(chub_env *e) {
    int rc = host_path_init(e);
    if (rc || e->h_bits) return rc;
    const size_t N = (size_t) e->hp.n_envs, W = (size_t) ((e->hp.S[0] + e->hp.S[1] + 63) / 64);
    // bits and tail in ONE pinned block and one device block: a caller that fills the handle's staging goes up in a single DMA
    HIP_TRY(hipHostMalloc((void **) &e->h_bits, N * W * sizeof(uint64_t) + N * 2 * sizeof(float), hipHostMallocDefault));
    e->h_tail = (float *) (e->h_bits + N * W);
    HIP_TRY(hipMalloc((void **) &e->d_bits, N * W * sizeof(uint64_t) + N * 2 * sizeof(float)));
    e->d_tail = (float *) (e->d_bits + N * W);
    return CHUB_OK;
}

int chub_host_bits(chub_env *e, uint64_t **bits_out, float **tail_out) {
    if (!e || !bits_out || !tail_out) return fail(CHUB_ERR_ARG, "null argument");
    HIP_TRY(hipSetDevice(e->device));
    int rc = bits_path_init(e);
    if (rc) return rc;
    *bits_out = e->h_bits;
    *tail_out = e->h_tail;
    return CHUB_OK;
}

// dense outputs (d_reward, d_done) or the packed block (d_packed: obs at stride D + 2, reward and done as floats behind each row)
static int step_bits_device(chub_env *e, const uint64_t *d_pile_bits, const float *d_tail, const double *d_exo_z, float *d_obs,
                            float *d_reward, uint8_t *d_done, float *d_packed, void *stream) {
    if (e->tick == 0) return fail(CHUB_ERR_ARG, "step() before reset()");
    if (e->tape_only && !e->tape_pk)
        return fail(CHUB_ERR_ARG, "a handle with registered tape classes steps through chub_step_tape / chub_step_tape_env only (its class "
                                  "rows hold the caller's arrival SoCs: cars admitted by this build's own draws would be given them)");
    HIP_TRY(hipSetDevice(e->device));
    const int D = e->hp.obs_dim;
    const float *rows = nullptr;
    if (e->hp.packed && e->hp.rng_mode == CHUB_RNG_PHILOX && !e->tape_pk) {
        // the production kernel reads the bits themselves: 8 bytes per env and word instead of a row of floats (and no expansion)
        e->cur_bits = d_pile_bits;
        e->cur_tail = d_tail;
    } else {
        launch_expand_bits(e->hp, d_pile_bits, d_tail, e->d_actions, (hipStream_t) stream);
        HIP_TRY(hipGetLastError());
        rows = e->d_actions;
    }
    const int rc = d_packed ? step_common(e, rows, d_exo_z, d_packed, D + 2, d_packed + D, D + 2, nullptr, d_packed + D + 1, stream)
                            : step_common(e, rows, d_exo_z, d_obs, D, d_reward, 1, d_done, nullptr, stream);
    e->cur_bits = nullptr;
    e->cur_tail = nullptr;
    return rc;
}

int chub_step_bits_device(chub_env *e, const uint64_t *d_pile_bits, const float *d_tail, const double *d_exo_z, float *d_obs,
                          float *d_reward, uint8_t *d_done, void *stream) {
    if (!e || !d_pile_bits || !d_tail || !d_obs || !d_reward || !d_done) return fail(CHUB_ERR_ARG, "null argument");
    return step_bits_device(e, d_pile_bits, d_tail, d_exo_z, d_obs, d_reward, d_done, nullptr, stream);
}

int chub_step_bits_device_packed(chub_env *e, const uint64_t *d_pile_bits, const float *d_tail, const double *d_exo_z, float *d_packed,
                                 void *stream) {
    if (!e || !d_pile_bits || !d_tail || !d_packed) return fail(CHUB_ERR_ARG, "null argument");
    return step_bits_device(e, d_pile_bits, d_tail, d_exo_z, nullptr, nullptr, nullptr, d_packed, stream);
}

int chub_step_bits(chub_env *e, const uint64_t *pile_bits, const float *tail, const double *exo_z, float *obs, float *reward,
                   uint8_t *done) {
    if (!e || !pile_bits || !tail || !obs || !reward || !done) return fail(CHUB_ERR_ARG, "null argument");
    HIP_TRY(hipSetDevice(e->device));
    int rc = bits_path_init(e);
    if (rc) return rc;
    const size_t N = (size_t) e->hp.n_envs, D = (size_t) e->hp.obs_dim, W = (size_t) ((e->hp.S[0] + e->hp.S[1] + 63) / 64);
    hipStream_t s = e->host_stream;
    if (e->hp.rng_mode == CHUB_RNG_COMPAT) {
        if (!exo_z) return fail(CHUB_ERR_ARG, "COMPAT mode needs exo_z");
        HIP_TRY(hipMemcpyAsync(e->d_exo_z, exo_z, N * 3 * sizeof(double), hipMemcpyHostToDevice, s));
    }
    // from the handle's pinned block (chub_host_bits) this is ONE DMA; any other host pointers are staged by the HIP runtime
    if (pile_bits == e->h_bits && tail == e->h_tail) {
        HIP_TRY(hipMemcpyAsync(e->d_bits, pile_bits, N * W * sizeof(uint64_t) + N * 2 * sizeof(float), hipMemcpyHostToDevice, s));
    } else {
        HIP_TRY(hipMemcpyAsync(e->d_bits, pile_bits, N * W * sizeof(uint64_t), hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemcpyAsync(e->d_tail, tail, N * 2 * sizeof(float), hipMemcpyHostToDevice, s));
    }
    void *v_obs = nullptr, *v_rew = nullptr, *v_done = nullptr;
    const bool direct = device_view(obs, &v_obs) && device_view(reward, &v_rew) && device_view(done, &v_done);
    if (direct) {
        rc = chub_step_bits_device(e, e->d_bits, e->d_tail, e->d_exo_z, (float *) v_obs, (float *) v_rew, (uint8_t *) v_done, s);
        if (rc) return rc;
    } else {
        rc = chub_step_bits_device(e, e->d_bits, e->d_tail, e->d_exo_z, e->d_obs, e->d_reward, e->d_done, s);
        if (rc) return rc;
        HIP_TRY(hipMemcpyAsync(obs, e->d_obs, N * D * sizeof(float), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipMemcpyAsync(reward, e->d_reward, N * sizeof(float), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipMemcpyAsync(done, e->d_done, N, hipMemcpyDeviceToHost, s));
    }
    HIP_TRY(hipStreamSynchronize(s));
    return CHUB_OK;
}

// ---- hipGraph capture of whole episodes (launch-bound loops: small shards, multi-GPU strong scaling) -----------------------
struct chub_graph {
    hipGraph_t graph;
    hipGraphExec_t exec;
    int device;
    chub_env *env;
    uint32_t ticks;       // resets + steps one replay covers
    uint32_t arg0;        // host tick - tick base at chub_graph_begin: the captured launches carry arg0 + 1 .. arg0 + ticks
    int t_begin, pc_begin;  // the handle's clocks at chub_graph_begin: what a replay starts from (baked into the launches)
    bool per_env_begin, per_env_end, predrawn_end;  // captured on per-env clocks (device state: nothing of them is baked in)
    std::vector<void *> masks;        // device masks of the captured calls on subsets of the envs
    std::vector<uint32_t> rel;        // [N] or empty: launch number within the graph of every env's last masked launch
    uint32_t full_rel;                // launch number of the graph's last launch on every env
    int t_end, pc_end;    // the handle's clocks after a replay
};

int chub_graph_begin(chub_env *e, void *stream) {
    if (!e || !stream) return fail(CHUB_ERR_ARG, "chub_graph_begin needs a handle and a created (non-default) stream");
    if (e->capturing) return fail(CHUB_ERR_ARG, "a capture is already in progress on this handle");
    if (e->hp.rng_mode != CHUB_RNG_PHILOX) return fail(CHUB_ERR_ARG, "graphs replay PHILOX steps (COMPAT takes host draws every step)");
    if (e->prof_on) return fail(CHUB_ERR_ARG, "per-kernel profiling is on");
    HIP_TRY(hipSetDevice(e->device));
    int rc = sync_ctx(e, (hipStream_t) stream);
    if (rc) return rc;
    HIP_TRY(hipStreamBeginCapture((hipStream_t) stream, hipStreamCaptureModeRelaxed));
    e->capturing = true;
    e->graph_tick0 = e->tick;
    e->graph_t0 = e->t;
    e->graph_pc0 = e->price_count;
    e->graph_predrawn0 = e->predrawn;
    e->graph_per_env0 = e->per_env;
    e->graph_full_tick0 = e->full_tick;
    e->graph_h_tick0 = e->h_tick;
    e->cap_masks.clear();
    e->cap_rel.clear();
    e->cap_full_rel = 0;
    e->cap_comm = nullptr;
    return CHUB_OK;
}

int chub_graph_end(chub_env *e, void *stream, chub_graph **out) {
    if (!e || !stream || !out) return fail(CHUB_ERR_ARG, "null argument");
    if (!e->capturing) return fail(CHUB_ERR_ARG, "no capture in progress");
    *out = nullptr;
    e->capturing = false;
    const uint32_t ticks = e->tick - e->graph_tick0;
    const int t_end = e->t, pc_end = e->price_count;
    const bool per_env_end = e->per_env, predrawn_end = e->predrawn;
    std::vector<void *> masks;
    masks.swap(e->cap_masks);
    auto drop_masks = [&]() {
        for (void *m : masks) (void) hipFree(m);
    };
    // nothing ran: the handle is where it was at chub_graph_begin; every chub_graph_launch moves it on by one replay
    e->tick = e->graph_tick0;
    e->t = e->graph_t0;
    e->price_count = e->graph_pc0;
    e->predrawn = e->graph_predrawn0;
    e->per_env = e->graph_per_env0;
    e->full_tick = e->graph_full_tick0;
    e->h_tick = e->graph_h_tick0;
    // overlapped gathers ran on the communicator's stream: it joins the capture's stream here (a capture ends on one stream; and the
    // next replay's first kernels overwrite the blocks the last gathers send)
    if (e->cap_comm) (void) chub_comm_join(e->cap_comm, stream);
    e->cap_comm = nullptr;
    // every replay moves the Philox tick base on by the ticks the graph covers (its last node)
    launch_tick_advance(e->d_tick_base, ticks, (hipStream_t) stream);
    hipGraph_t g = nullptr;
    hipError_t he = hipStreamEndCapture((hipStream_t) stream, &g);
    if (he != hipSuccess || !g) {
        drop_masks();
        return fail(CHUB_ERR_HIP, std::string("hipStreamEndCapture: ") + hipGetErrorString(he));
    }
    if (ticks == 0 || (ticks & 1u)) {
        (void) hipGraphDestroy(g);
        drop_masks();
        return fail(CHUB_ERR_ARG, "a graph must cover an even, non-zero number of resets + steps (double-buffered draws)");
    }
    hipGraphExec_t x = nullptr;
    he = hipGraphInstantiate(&x, g, nullptr, nullptr, 0);
    if (he != hipSuccess) {
        (void) hipGraphDestroy(g);
        drop_masks();
        return fail(CHUB_ERR_HIP, std::string("hipGraphInstantiate: ") + hipGetErrorString(he));
    }
    chub_graph *cg = new chub_graph();
    cg->graph = g;
    cg->exec = x;
    cg->device = e->device;
    cg->env = e;
    cg->ticks = ticks;
    cg->arg0 = e->graph_tick0 - e->graph_base;
    cg->t_begin = e->graph_t0;
    cg->pc_begin = e->graph_pc0;
    cg->per_env_begin = e->graph_per_env0;
    cg->per_env_end = per_env_end;
    cg->predrawn_end = predrawn_end;
    cg->masks.swap(masks);
    cg->rel.swap(e->cap_rel);
    cg->full_rel = e->cap_full_rel;
    cg->t_end = t_end;
    cg->pc_end = pc_end;
    *out = cg;
    return CHUB_OK;
}

int chub_graph_launch(chub_graph *g, void *stream) {
    if (!g) return fail(CHUB_ERR_ARG, "null graph");
    HIP_TRY(hipSetDevice(g->device));
    chub_env *e = g->env;
    if (e->capturing) return fail(CHUB_ERR_ARG, "a capture is in progress on this handle");
    if (g->per_env_begin != e->per_env)
        return fail(CHUB_ERR_ARG, g->per_env_begin ? "this graph was captured on per-env clocks: the handle is in lock-step (make a call on a subset of the envs first)"
                                                   : "this graph replays lock-step calls: the envs of this handle run on their own clocks (reset all of them first)");
    // a lock-step replay repeats the clocks of its launches verbatim (slot of day, price-noise phase: kernel arguments): it
    // continues the handle's run only from where the capture started.  On per-env clocks the clocks are device state and a replay
    // continues from wherever every env is.
    if (!g->per_env_begin && (e->t != g->t_begin || ((e->price_count ^ g->pc_begin) & 3) != 0))
        return fail(CHUB_ERR_ARG, "the handle is not at the clock this graph was captured at (slot of day " + std::to_string(g->t_begin) +
                                      ", steps since reset mod 4 = " + std::to_string(g->pc_begin & 3) + "): a replay bakes the clocks in");
    // The captured launches carry the arguments arg0 + 1 .. arg0 + ticks, and a launch's effective Philox tick is its argument
    // + the device-side tick base.  Calls issued one by one since the capture (or since the last replay) have moved the host
    // tick on without moving the base: bring the base to where argument arg0 + k means tick + k again, so that a replay never
    // runs on a tick an eager call has already used (nor the other way round).
    const uint32_t delta = (e->tick - e->graph_base) - g->arg0;
    if (delta != 0u) {
        // The per-env clocks are double-buffered by the parity of that same argument: the graph's first launch reads buffer
        // (arg0 + 1) & 1, while the live clocks -- after an odd number of eager calls -- sit in buffer (arg0 + delta + 1) & 1,
        // the other one.  Bring them over before the base moves (every env would otherwise replay from a stale slot
        // of day and price phase).
        if (g->per_env_begin && (delta & 1u)) {
            const size_t N = (size_t) e->hp.n_envs;
            launch_keep_clocks(e->d_env_clk + (size_t) ((g->arg0 + 1u) & 1u) * N, e->d_env_clk + (size_t) (g->arg0 & 1u) * N, (int64_t) N,
                               (hipStream_t) stream);
        }
        launch_tick_advance(e->d_tick_base, delta, (hipStream_t) stream);
        HIP_TRY(hipGetLastError());
        e->graph_base += delta;
    }
    HIP_TRY(hipGraphLaunch(g->exec, (hipStream_t) stream));
    // the replay covers g->ticks resets + steps: its last node moves the device-side tick base on, the host mirrors it, so that
    // calls issued one by one afterwards continue the same tick sequence (their argument is the tick minus the base)
    const uint32_t tick0 = e->tick;
    e->tick += g->ticks;
    e->graph_base += g->ticks;
    e->per_env = g->per_env_end;
    if (!g->per_env_begin || !g->per_env_end) {  // the replay ran on, or ended on, the one clock of the handle
        e->t = g->t_end;
        e->price_count = g->pc_end;
    }
    e->predrawn = g->predrawn_end;
    if (g->full_rel) e->full_tick = tick0 + g->full_rel;
    if (!g->rel.empty()) {
        const size_t N = (size_t) e->hp.n_envs;
        if (e->h_tick.size() != N) e->h_tick.assign(N, 0u);
        for (size_t i = 0; i < N; i++)
            if (g->rel[i]) e->h_tick[i] = tick0 + g->rel[i];
    }
    return CHUB_OK;
}

int chub_graph_destroy(chub_graph *g) {
    if (!g) return CHUB_OK;
    (void) hipSetDevice(g->device);
    (void) hipGraphExecDestroy(g->exec);
    (void) hipGraphDestroy(g->graph);
    (void) hipDeviceSynchronize();  // a replay still in flight reads the masks
    for (void *m : g->masks) (void) hipFree(m);
    delete g;
    return CHUB_OK;
}

// ---- device memory and streams for hosts without a GPU array library (the Python host is ctypes + numpy) ----------------
int chub_malloc_device(int device, int64_t bytes, void **out) {
    if (!out || bytes <= 0) return fail(CHUB_ERR_ARG, "bad argument");
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipMalloc(out, (size_t) bytes));
    return CHUB_OK;
}
int chub_free_device(int device, void *p) {
    if (!p) return CHUB_OK;
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipFree(p));
    return CHUB_OK;
}
int chub_copy_to_host(int device, void *dst, const void *d_src, int64_t bytes, void *stream) {
    if (!dst || !d_src || bytes < 0) return fail(CHUB_ERR_ARG, "bad argument");
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipMemcpyAsync(dst, d_src, (size_t) bytes, hipMemcpyDeviceToHost, (hipStream_t) stream));
    HIP_TRY(hipStreamSynchronize((hipStream_t) stream));
    return CHUB_OK;
}
int chub_copy_to_device(int device, void *d_dst, const void *src, int64_t bytes, void *stream) {
    if (!d_dst || !src || bytes < 0) return fail(CHUB_ERR_ARG, "bad argument");
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipMemcpyAsync(d_dst, src, (size_t) bytes, hipMemcpyHostToDevice, (hipStream_t) stream));
    HIP_TRY(hipStreamSynchronize((hipStream_t) stream));
    return CHUB_OK;
}
int chub_alloc_host(int device, int64_t bytes, void **out) {
    if (!out || bytes <= 0) return fail(CHUB_ERR_ARG, "bad argument");
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipHostMalloc(out, (size_t) bytes, hipHostMallocDefault));
    return CHUB_OK;
}
int chub_free_host(int device, void *p) {
    if (!p) return CHUB_OK;
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipHostFree(p));
    return CHUB_OK;
}
int chub_stream_create(int device, void **out) {
    if (!out) return fail(CHUB_ERR_ARG, "null argument");
    HIP_TRY(hipSetDevice(device));
    hipStream_t s;
    HIP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *out = (void *) s;
    return CHUB_OK;
}
int chub_stream_destroy(int device, void *stream) {
    if (!stream) return CHUB_OK;
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipStreamDestroy((hipStream_t) stream));
    return CHUB_OK;
}
int chub_stream_sync(int device, void *stream) {
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipStreamSynchronize((hipStream_t) stream));
    return CHUB_OK;
}

// ---- tape mode (parity instrument, PHILOX handles): recorded decisions replayed through the production kernels -------
int chub_tape_register_soc(chub_env *e, const float *soc, int32_t count, uint32_t *class_ids) {
    if (!e || !soc || !class_ids || count < 0) return fail(CHUB_ERR_ARG, "bad argument");
    if (e->hp.rng_mode != CHUB_RNG_PHILOX) return fail(CHUB_ERR_ARG, "tape mode needs a PHILOX handle");
    if (e->tape_classes + count > kSocLevels) return fail(CHUB_ERR_UNSUPPORTED, "too many tape classes (chub_tape_clear_soc starts over)");
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipDeviceSynchronize());
    // The state word has room for kSocLevels classes and the production kernel is the one to be replayed, so the caller's arrival
    // SoCs take the PLACE of this build's own classes, first come first row: from here on the handle is a tape handle (cars that a
    // Philox step or reset admits would be given the caller's SoCs)
    const bool cp = e->hp.constant_charging != 0;
    std::vector<float> rows((size_t) count * kClsRow * 2);
    const size_t first = (size_t) e->tape_classes;
    for (int k = 0; k < 2; k++) {  // (checked for the whole batch before anything is overwritten)
        const bool fast = e->hp.type[k] == CHUB_FAST;
        float tt_max = 0.0f;
        for (float t : e->h_ttab[k]) tt_max = t > tt_max ? t : tt_max;
        for (int i = 0; i < count; i++) {
            // stay_time = ceil(soc_to_time(target) - soc_to_time(soc)) + late (late <= 15) must fit the state word's 5-bit fields, as
            // chub_create checks for the build's own classes: an arrival SoC whose stay could not is refused, not clamped
            float row[kClsRow * 2];
            build_class_row(fast, cp, e->hp.cc, soc[i], row);
            if (!(soc[i] >= 0.0f && soc[i] <= 100.0f) || (int) ceilf(tt_max - row[1]) + 15 > 31)
                return fail(CHUB_ERR_ARG, "chub_tape_register_soc: an arrival SoC whose stay could exceed 31 slots (or outside 0 .. 100)");
        }
    }
    e->tape_only = e->tape_only || count > 0;
    for (int k = 0; k < 2; k++) {
        const bool fast = e->hp.type[k] == CHUB_FAST;
        for (int i = 0; i < count; i++) build_class_row(fast, cp, e->hp.cc, soc[i], &rows[(size_t) i * kClsRow * 2]);
        if (count) {
            HIP_TRY(hipMemcpy((void *) (e->tb.cls[k] + first * kClsRow * 2), rows.data(), rows.size() * sizeof(float), hipMemcpyHostToDevice));
            HIP_TRY(hipMemcpy((void *) (e->tb.cls_soc0[k] + first), soc, (size_t) count * sizeof(float), hipMemcpyHostToDevice));
            memcpy(&e->h_cls[k][first * kClsRow * 2], rows.data(), rows.size() * sizeof(float));
            memcpy(&e->h_soc0[k][first], soc, (size_t) count * sizeof(float));
        }
    }
    for (int i = 0; i < count; i++) class_ids[i] = (uint32_t) (first + (size_t) i);
    e->tape_classes += count;
    return CHUB_OK;
}

int chub_tape_clear_soc(chub_env *e) {
    if (!e) return fail(CHUB_ERR_ARG, "null handle");
    // the next chub_tape_register_soc starts at class 0 again.  Cars of the old classes still in their slots would be re-read against the
    // new rows: the caller clears between episodes, right in front of the chub_reset_tape that wipes every slot (evs_reset, CHS.hpp:1209-1231),
    // and chub_step_tape refuses to run until that reset has happened
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipDeviceSynchronize());
    e->tape_classes = 0;
    e->tape_stale = true;
    return CHUB_OK;
}

int chub_set_slots(chub_env *e, const int32_t *rows) {
    if (!e || !rows) return fail(CHUB_ERR_ARG, "null argument");
    if (e->hp.rng_mode != CHUB_RNG_PHILOX) return fail(CHUB_ERR_ARG, "chub_set_slots needs a PHILOX handle");
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipDeviceSynchronize());
    const HubParams &hp = e->hp;
    const size_t N = (size_t) hp.n_envs, S = (size_t) (hp.S[0] + hp.S[1]);
    const uint32_t n_classes = (uint32_t) kSocLevels;
    std::vector<uint32_t> st(N * S, 0u);
    std::vector<uint8_t> stay(N * S, 0);
    for (size_t env = 0; env < N; env++)
        for (int k = 0; k < 2; k++)
            for (size_t i = 0; i < (size_t) hp.S[k]; i++) {
                const int32_t *r = rows + (env * S + (k ? (size_t) hp.S[0] : 0) + i) * 6;
                if (r[0] < 0) continue;  // empty slot
                const int left = r[2] - r[3];
                if ((uint32_t) r[0] >= n_classes || r[1] < 0 || r[1] >= kLevels || r[2] < 1 || r[2] > 31 || left < 1 || r[4] < 0 ||
                    r[4] >= kClsRow)
                    return fail(CHUB_ERR_ARG, "chub_set_slots: field out of range");
                const size_t idx = env * S + (k ? (size_t) hp.S[0] : 0) + i;  // PHILOX state is hub-major
                // the state word of chub_kernels.hip: what is left of the stay, charging flag, car_steps taken, class, target level
                st[idx] = (uint32_t) left | (r[5] ? 32u : 0u) | ((uint32_t) r[4] << 6) | ((uint32_t) r[0] << 11) | ((uint32_t) r[1] << 22);
                stay[idx] = (uint8_t) r[2];
            }
    HIP_TRY(hipMemcpy(e->sl.hot, st.data(), st.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(e->sl.stay8, stay.data(), stay.size(), hipMemcpyHostToDevice));
    return CHUB_OK;
}

int chub_set_station_queue(chub_env *e, const int32_t *line) {
    if (!e || !line) return fail(CHUB_ERR_ARG, "null argument");
    e->walked_tick = 0;
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipDeviceSynchronize());
    const size_t N = (size_t) e->hp.n_envs;
    std::vector<uint32_t> rec;
    int rc = fetch(rec, (const uint32_t *) e->st.rec, 8 * N);
    if (rc) return rc;
    for (size_t env = 0; env < N; env++)
        for (int k = 0; k < 2; k++) {
            if (line[env * 2 + k] < 0 || line[env * 2 + k] > kMaxLine) return fail(CHUB_ERR_ARG, "queue length out of range");
            uint32_t &w = rec[4 * ((size_t) k * N + env) + 3];
            w = (w & ~15u) | (uint32_t) line[env * 2 + k];
        }
    HIP_TRY(hipMemcpy(e->st.rec, rec.data(), rec.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    // the next step's station draws were decoded one launch ahead against the queue lengths just overwritten: the next launch
    // draws and decodes its own (k_draw_levels; same Philox counters)
    e->predrawn = false;
    return CHUB_OK;
}

int chub_step_tape(chub_env *e, const float *actions, const uint64_t *pk_tape, const uint32_t *car_tape, float *obs, float *reward,
                   uint8_t *done) {
    return chub_step_tape_env(e, actions, pk_tape, car_tape, nullptr, nullptr, 0, obs, reward, done);
}

// ... the whole step from the tape: the per-env tail (k_env<.., TAPE> / the tail half of k_step_fused<.., TAPE>) takes the step's exogenous
// normals and the forecourt's arrivals from the caller as well
int chub_step_tape_env(chub_env *e, const float *actions, const uint64_t *pk_tape, const uint32_t *car_tape, const double *exo_z,
                       const uint32_t *hv_tape, int32_t hv_w, float *obs, float *reward, uint8_t *done) {
    if (!e || !actions || !pk_tape || !car_tape || !obs || !reward || !done) return fail(CHUB_ERR_ARG, "null argument");
    if (e->hp.rng_mode != CHUB_RNG_PHILOX || !e->hp.packed)
        return fail(CHUB_ERR_ARG, "tape mode drives the packed PHILOX slot kernel: the hub shape must be one it covers");
    if ((exo_z != nullptr) != (hv_tape != nullptr) || (hv_tape && hv_w < 1)) return fail(CHUB_ERR_ARG, "the tail's tape is exo_z [N][3] AND hv_tape [N][hv_w >= 1]");
    if (e->tape_stale) return fail(CHUB_ERR_ARG, "chub_tape_clear_soc was called: reset (chub_reset_tape) before the next tape step");
    HIP_TRY(hipSetDevice(e->device));
    const size_t N = (size_t) e->hp.n_envs, S = (size_t) (e->hp.S[0] + e->hp.S[1]);
    uint32_t *d_hv = nullptr;
    if (hv_tape) {
        // at most as many arrivals per step as the handle's forecourt can see from its own tables (the waiting list's explicit entries
        // are sized for that, chub_create), each with its SoC on the tape
        const uint32_t most = (uint32_t) ((e->hp.qcap + 1) / 2);
        for (size_t i = 0; i < N; i++)
            if (hv_tape[i * (size_t) hv_w] > most || hv_tape[i * (size_t) hv_w] > (uint32_t) (hv_w - 1))
                return fail(CHUB_ERR_ARG, "hv_tape: more FCEV arrivals in one step than the handle's forecourt (or the tape's width) takes");
        HIP_TRY(hipMalloc((void **) &d_hv, N * (size_t) hv_w * sizeof(uint32_t)));
        hipError_t he1 = hipMemcpy(d_hv, hv_tape, N * (size_t) hv_w * sizeof(uint32_t), hipMemcpyHostToDevice);
        if (he1 == hipSuccess) he1 = hipMemcpy(e->d_exo_z, exo_z, N * 3 * sizeof(double), hipMemcpyHostToDevice);
        if (he1 != hipSuccess) {
            (void) hipFree(d_hv);
            return fail(CHUB_ERR_HIP, std::string("hipMemcpy: ") + hipGetErrorString(he1));
        }
    }
    // the car tape is in hub order [N][S][2] (station 0's slots first), which is the kernel's own slot order
    const std::vector<uint32_t> ct(car_tape, car_tape + 2 * N * S);
    uint64_t *d_pk = nullptr;
    uint32_t *d_ct = nullptr;
    hipError_t he = hipMalloc((void **) &d_pk, 2 * N * sizeof(uint64_t));
    if (he == hipSuccess) he = hipMalloc((void **) &d_ct, ct.size() * sizeof(uint32_t));
    if (he != hipSuccess) {
        (void) hipFree(d_pk);
        (void) hipFree(d_hv);
        return fail(CHUB_ERR_HIP, std::string("hipMalloc: ") + hipGetErrorString(he));
    }
    int rc = CHUB_OK;
    auto done_ = [&](int code) {
        (void) hipDeviceSynchronize();
        (void) hipFree(d_pk);
        (void) hipFree(d_ct);
        (void) hipFree(d_hv);
        return code;
    };
    if (hipMemcpy(d_pk, pk_tape, 2 * N * sizeof(uint64_t), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(d_ct, ct.data(), ct.size() * sizeof(uint32_t), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(e->d_actions, actions, N * (size_t) e->hp.act_dim * sizeof(float), hipMemcpyHostToDevice) != hipSuccess)
        return done_(fail(CHUB_ERR_HIP, "hipMemcpy failed"));
    e->tape_pk = d_pk;
    e->tape_car = d_ct;
    e->tape_hv = d_hv;
    e->tape_hv_w = hv_w;
    e->tape_tail = d_hv != nullptr;
    rc = chub_step_device(e, e->d_actions, d_hv ? e->d_exo_z : nullptr, e->d_obs, e->d_reward, e->d_done, nullptr);
    e->tape_pk = nullptr;
    e->tape_car = nullptr;
    e->tape_hv = nullptr;
    e->tape_hv_w = 0;
    e->tape_tail = false;
    if (rc) return done_(rc);
    if (hipMemcpy(obs, e->d_obs, N * (size_t) e->hp.obs_dim * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(reward, e->d_reward, N * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(done, e->d_done, N, hipMemcpyDeviceToHost) != hipSuccess)
        return done_(fail(CHUB_ERR_HIP, "hipMemcpy failed"));
    return done_(CHUB_OK);
}

int chub_reset_tape(chub_env *e, const uint32_t *occ_tape, const uint32_t *car_tape, float *obs) {
    return chub_reset_tape_env(e, occ_tape, car_tape, nullptr, nullptr, obs);
}

// ... the whole reset from the tape: the tail (k_env<RESET, .., TAPE>) takes renew_reset's days (REN:51-53) and make_state's normals
// (MGR:344-361) from the caller
int chub_reset_tape_env(chub_env *e, const uint32_t *occ_tape, const uint32_t *car_tape, const int32_t *exo_days, const double *exo_z, float *obs) {
    if (!e || !occ_tape || !car_tape || !obs) return fail(CHUB_ERR_ARG, "null argument");
    if ((exo_days != nullptr) != (exo_z != nullptr)) return fail(CHUB_ERR_ARG, "the tail's tape of a reset is exo_days [N][2] AND exo_z [N][3]");
    if (exo_days)
        for (size_t i = 0; i < (size_t) e->hp.n_envs; i++)
            if (exo_days[2 * i] < 0 || exo_days[2 * i] >= 100 || exo_days[2 * i + 1] < 0 || exo_days[2 * i + 1] >= 150)
                return fail(CHUB_ERR_ARG, "exo_days out of range");
    if (e->hp.rng_mode != CHUB_RNG_PHILOX || !e->hp.packed)
        return fail(CHUB_ERR_ARG, "tape mode drives the packed PHILOX slot kernel: the hub shape must be one it covers");
    if (e->capturing) return fail(CHUB_ERR_ARG, "tape mode cannot be captured");
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipDeviceSynchronize());
    const size_t N = (size_t) e->hp.n_envs, S = (size_t) (e->hp.S[0] + e->hp.S[1]);
    uint32_t *d_ct = nullptr;
    HIP_TRY(hipMalloc((void **) &d_ct, 2 * N * S * sizeof(uint32_t)));
    auto done_ = [&](int code) {
        (void) hipDeviceSynchronize();
        (void) hipFree(d_ct);
        return code;
    };
    // the reset is the launch with argument tick + 1 - base: its slot kernel reads the units' occupancy draws from pk[that & 1],
    // where k_reset_levels would have left this build's own
    const uint32_t arg = e->tick + 1u - e->graph_base;
    if (hipMemcpy((void *) e->st.pk[arg & 1u], occ_tape, 2 * N * sizeof(uint32_t), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(d_ct, car_tape, 2 * N * S * sizeof(uint32_t), hipMemcpyHostToDevice) != hipSuccess)
        return done_(fail(CHUB_ERR_HIP, "hipMemcpy failed"));
    if (exo_days && (hipMemcpy(e->d_exo_days, exo_days, 2 * N * sizeof(int32_t), hipMemcpyHostToDevice) != hipSuccess ||
                     hipMemcpy(e->d_exo_z, exo_z, 3 * N * sizeof(double), hipMemcpyHostToDevice) != hipSuccess))
        return done_(fail(CHUB_ERR_HIP, "hipMemcpy failed"));
    e->tape_car = d_ct;
    e->tape_tail = exo_days != nullptr;
    int rc = chub_reset_device(e, exo_days ? e->d_exo_days : nullptr, exo_days ? e->d_exo_z : nullptr, e->d_obs, nullptr);
    e->tape_car = nullptr;
    e->tape_tail = false;
    if (rc) return done_(rc);
    e->tape_stale = false;  // evs_reset has wiped every slot
    if (hipMemcpy(obs, e->d_obs, N * (size_t) e->hp.obs_dim * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess)
        return done_(fail(CHUB_ERR_HIP, "hipMemcpy failed"));
    return done_(CHUB_OK);
}

int chub_random_actions_device(chub_env *e, uint64_t key, uint32_t batch, float *d_actions, void *stream) {
    if (!e || !d_actions) return fail(CHUB_ERR_ARG, "null argument");
    HIP_TRY(hipSetDevice(e->device));
    launch_random_actions(e->hp, key, batch, d_actions, (hipStream_t) stream);
    HIP_TRY(hipGetLastError());
    return CHUB_OK;
}

// ------------------------------------------------------------------------------- introspection
int chub_get_slots(chub_env *e, float *out) {
    if (!e || !out) return fail(CHUB_ERR_ARG, "null argument");
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipDeviceSynchronize());
    const HubParams &hp = e->hp;
    const size_t N = (size_t) hp.n_envs, S = (size_t) (hp.S[0] + hp.S[1]), NS = N * S;
    const bool philox = hp.rng_mode == CHUB_RNG_PHILOX;
    std::vector<float> soc;
    const std::vector<float> *cls = e->h_cls, *soc0 = e->h_soc0, *ttab = e->h_ttab;
    std::vector<uint32_t> hot;
    int rc;
    if ((rc = sync_ctx(e, nullptr))) return rc;
    {   // current SoC: the arrival SoC advanced by the recorded number of car_steps, on the device (k_replay_soc)
        float *d_soc = nullptr;
        HIP_TRY(hipMalloc((void **) &d_soc, NS * sizeof(float)));
        launch_replay_soc(hp, e->d_ctx, d_soc, nullptr);
        rc = fetch(soc, (const float *) d_soc, NS);
        (void) hipFree(d_soc);
        if (rc) return rc;
    }
    std::vector<uint8_t> stay8;
    if ((rc = fetch(hot, (const uint32_t *) e->sl.hot, (philox ? 1 : 4) * NS))) return rc;
    if (philox && (rc = fetch(stay8, (const uint8_t *) e->sl.stay8, NS))) return rc;
    for (size_t env = 0; env < N; env++) {
        float *o = out + env * 9 * S;
        for (int k = 0; k < 2; k++) {
            const size_t n = (size_t) hp.S[k];
            for (size_t i = 0; i < n; i++) {
                const size_t idx = philox ? env * S + (k ? (size_t) hp.S[0] : 0) + i : (size_t) hp.base[k] + env * n + i;
                float power = 0, t_target = 0, t_soc = 0, arrive = 0;
                int left, stay, lev;
                bool chg;
                if (philox) {  // 4-byte state: everything else comes from the class row and the table of target times (see chub_kernels.hip)
                    const uint32_t w0 = hot[idx];
                    left = (int) (w0 & 31u); chg = (w0 & 32u) != 0; stay = (int) stay8[idx]; lev = (int) (w0 >> 22);
                    if (left > 0) {
                        const size_t c = (size_t) ((w0 >> 11) & 2047u), at = (c * kClsRow + ((w0 >> 6) & 31u)) * 2;
                        power = cls[k][at]; t_soc = cls[k][at + 1]; arrive = soc0[k][c];
                        t_target = ttab[k][(size_t) lev];
                    }
                } else {
                    memcpy(&power, &hot[4 * idx + 0], 4);
                    memcpy(&arrive, &hot[4 * idx + 1], 4);  // (the record keeps the arrival SoC; the target's time is the level's table entry)
                    memcpy(&t_soc, &hot[4 * idx + 2], 4);
                    const uint32_t w = hot[4 * idx + 3];
                    left = (int) (w & 127u); chg = (w & 128u) != 0; stay = (int) ((w >> 8) & 127u); lev = (int) ((w >> 15) & 1023u);
                    if (left > 0) t_target = ttab[k][(size_t) lev];
                }
                const bool car = left > 0;
                const float tr = (float) lev / 999.0f;
                const float target = tr * (100.0f - 80.0f) + 80.0f;  // uniform_rand(80, 100) at level lev, CHS.hpp:35-44
                float em = 0.0f;
                if (car) {  // Station::situation["emergency"] as calculate_needed leaves it (CHS.hpp:879-898)
                    float need = t_target - t_soc;
                    if (need > 0) em = ((float) left <= ceilf(need)) ? 10.0f : (float) pow((double) (need / (float) left), 2);
                }
                o[0 * n + i] = car ? 1.0f : 0.0f;
                o[1 * n + i] = chg ? 1.0f : 0.0f;
                o[2 * n + i] = em;
                o[3 * n + i] = car ? power : 0.0f;
                o[4 * n + i] = car ? soc[idx] : 0.0f;
                o[5 * n + i] = car ? arrive : 0.0f;
                o[6 * n + i] = car ? target : 0.0f;
                o[7 * n + i] = car ? (float) stay : -1.0f;
                o[8 * n + i] = car ? (float) (stay - left) : -1.0f;
            }
            o += 9 * n;
        }
    }
    return CHUB_OK;
}

int chub_get_station_scalars(chub_env *e, double *out) {
    if (!e || !out) return fail(CHUB_ERR_ARG, "null argument");
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipDeviceSynchronize());
    const size_t N = (size_t) e->hp.n_envs;
    std::vector<uint32_t> rec;
    std::vector<uint16_t> clk;  // station_time_hole (CHS.hpp:1204) is the env's own slot of day
    int rc;
    if ((rc = fetch(rec, (const uint32_t *) e->st.rec, 8 * N)) || (rc = fetch_clocks(e, clk))) return rc;
    for (size_t env = 0; env < N; env++)
        for (int k = 0; k < 2; k++) {
            double *o = out + (env * 2 + k) * 8;
            const uint32_t *r = &rec[4 * ((size_t) k * N + env)];
            float mn, ch, mx;
            memcpy(&mn, &r[0], 4); memcpy(&ch, &r[1], 4); memcpy(&mx, &r[2], 4);
            o[0] = mn; o[1] = ch; o[2] = mx;
            o[3] = (double) pkd_cars(r[3]);  // car_number
            o[4] = (double) pkd_line(r[3]);  // line
            o[5] = (double) pkd_flow(r[3]);  // flow_in_number[-1] (can be negative right after reset)
            o[6] = (double) (clk[env] & 127u); o[7] = e->hp.transformer_limit[k];
        }
    return CHUB_OK;
}

int chub_set_telemetry(chub_env *e, int enabled) {
    if (!e) return fail(CHUB_ERR_ARG, "null handle");
    HIP_TRY(hipSetDevice(e->device));
    if (enabled && !e->h_telem && !e->d_telem) {
        const size_t N = (size_t) e->hp.n_envs, D = (size_t) e->hp.obs_dim;
        const size_t count = N * (size_t) kTelemCount + N * D + N;
        double *base = nullptr;
        if (N * (size_t) e->hp.act_dim * sizeof(float) <= 16384) {
            // handles of a few envs (the bound chub_step uses for reading the caller's arrays in place: the drop-in class): one block of
            // pinned host memory, mapped into the device's address space -- the tail kernel's telemetry stores cross PCIe as posted
            // writes while it runs, and reading them back is a host read: no device read per step
            HIP_TRY(hipHostMalloc((void **) &e->h_telem, count * sizeof(double), hipHostMallocDefault));
            memset(e->h_telem, 0, count * sizeof(double));
            void *dv = nullptr;
            if (!device_view(e->h_telem, &dv)) {
                (void) hipHostFree(e->h_telem);
                e->h_telem = nullptr;
                return fail(CHUB_ERR_HIP, "pinned telemetry block is not visible to the device");
            }
            base = (double *) dv;
        } else {
            // a batch: 27 MB per step at 65 536 envs would stall the tail on posted PCIe writes -- the block stays in HBM (outside the
            // arena: a snapshot does not carry it) and the getters copy
            HIP_TRY(hipMalloc((void **) &e->d_telem, count * sizeof(double)));
            hipError_t he = hipMemset(e->d_telem, 0, count * sizeof(double));
            if (he != hipSuccess) {
                (void) hipFree(e->d_telem);
                e->d_telem = nullptr;
                return fail(CHUB_ERR_HIP, std::string("hipMemset: ") + hipGetErrorString(he));
            }
            base = e->d_telem;
        }
        e->ev.telem = base;
        e->ev.obs64 = e->ev.telem + N * (size_t) kTelemCount;
        e->ev.reward64 = e->ev.obs64 + N * D;
    }
    e->hp.telemetry = enabled ? 1 : 0;
    e->ctx_dirty = true;
    return CHUB_OK;
}

static int need_telemetry(chub_env *e) {
    if (!e->h_telem && !e->d_telem) return fail(CHUB_ERR_ARG, "telemetry is off: call chub_set_telemetry(env, 1) first");
    return 0;
}

int chub_telemetry_host(chub_env *e, double **telem, double **obs64, double **reward64) {
    if (!e || !telem || !obs64 || !reward64) return fail(CHUB_ERR_ARG, "null argument");
    int rc = need_telemetry(e);
    if (rc) return rc;
    if (!e->h_telem)
        return fail(CHUB_ERR_UNSUPPORTED, "the telemetry block of a handle of this size lives in device memory: read it with chub_get_telemetry / "
                                          "chub_get_obs_f64 / chub_get_reward_f64");
    const size_t N = (size_t) e->hp.n_envs, D = (size_t) e->hp.obs_dim;
    *telem = e->h_telem;
    *obs64 = e->h_telem + N * (size_t) kTelemCount;
    *reward64 = *obs64 + N * D;
    return CHUB_OK;
}

// `count` doubles of the telemetry block from `offset` on, wherever the block lives (the work in flight is waited for)
static int telem_fetch(chub_env *e, size_t offset, size_t count, double *dst) {
    int rc = need_telemetry(e);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipDeviceSynchronize());
    if (e->h_telem) memcpy(dst, e->h_telem + offset, count * sizeof(double));
    else HIP_TRY(hipMemcpy(dst, e->d_telem + offset, count * sizeof(double), hipMemcpyDeviceToHost));
    return CHUB_OK;
}

int chub_get_telemetry(chub_env *e, double *out) {
    if (!e || !out) return fail(CHUB_ERR_ARG, "null argument");
    const size_t N = (size_t) e->hp.n_envs;
    std::vector<double> t(N * (size_t) kTelemCount);
    int rc = telem_fetch(e, 0, t.size(), t.data());
    if (rc) return rc;
    for (size_t env = 0; env < N; env++)
        for (int i = 0; i < kTelemCount; i++) out[env * kTelemCount + i] = t[(size_t) i * N + env];
    return CHUB_OK;
}

int chub_get_obs_f64(chub_env *e, double *out) {
    if (!e || !out) return fail(CHUB_ERR_ARG, "null argument");
    const size_t N = (size_t) e->hp.n_envs;
    return telem_fetch(e, N * (size_t) kTelemCount, N * (size_t) e->hp.obs_dim, out);
}

int chub_get_reward_f64(chub_env *e, double *out) {
    if (!e || !out) return fail(CHUB_ERR_ARG, "null argument");
    const size_t N = (size_t) e->hp.n_envs;
    return telem_fetch(e, N * (size_t) kTelemCount + N * (size_t) e->hp.obs_dim, N, out);
}

int chub_set_rng_compat_seeds(chub_env *e, const uint32_t *seeds) {
    if (!e || !seeds) return fail(CHUB_ERR_ARG, "null argument");
    e->walked_tick = 0;  // (a walk that ran ahead drew from the streams as they were)
    if (e->hp.rng_mode != CHUB_RNG_COMPAT) return fail(CHUB_ERR_ARG, "handle is not in COMPAT mode");
    HIP_TRY(hipSetDevice(e->device));
    const size_t N = (size_t) e->hp.n_envs;
    std::vector<uint32_t> g(N * 32), m(N);
    for (size_t i = 0; i < N; i++) {
        // glibc srandom_r, TYPE_3: LCG fill of 31 words, front = 3, rear = 0, 310 warm-up draws
        uint32_t *r = &g[i * 32];
        uint32_t sd = seeds[2 * i] ? seeds[2 * i] : 1u;
        int32_t word = (int32_t) sd;
        r[0] = (uint32_t) word;
        for (int j = 1; j < 31; j++) {
            long hi = word / 127773, lo = word % 127773;
            word = (int32_t) (16807 * lo - 2836 * hi);
            if (word < 0) word += 2147483647;
            r[j] = (uint32_t) word;
        }
        uint32_t f = 3, b = 0;
        for (int j = 0; j < 310; j++) {
            r[f] += r[b];
            f = (f + 1 == 31) ? 0 : f + 1;
            b = (b + 1 == 31) ? 0 : b + 1;
        }
        r[31] = f;
        uint32_t x = seeds[2 * i + 1] % 2147483647u;  // minstd_rand0::seed
        m[i] = x ? x : 1u;
    }
    HIP_TRY(hipMemcpy(e->cr.g3[e->rng_cur], g.data(), g.size() * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(e->cr.minstd3[e->rng_cur], m.data(), m.size() * 4, hipMemcpyHostToDevice));
    return CHUB_OK;
}

int chub_set_rng_compat_state(chub_env *e, const uint32_t *state) {
    if (!e || !state) return fail(CHUB_ERR_ARG, "null argument");
    e->walked_tick = 0;  // (a walk that ran ahead of its step is void)
    if (e->hp.rng_mode != CHUB_RNG_COMPAT) return fail(CHUB_ERR_ARG, "handle is not in COMPAT mode");
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipDeviceSynchronize());
    const size_t N = (size_t) e->hp.n_envs;
    std::vector<uint32_t> g(N * 32), m(N);
    for (size_t i = 0; i < N; i++) {
        if (state[i * 33 + 31] >= 31u) return fail(CHUB_ERR_ARG, "front index must be < 31");
        memcpy(&g[i * 32], &state[i * 33], 32 * 4);
        m[i] = state[i * 33 + 32];
    }
    HIP_TRY(hipMemcpy(e->cr.g3[e->rng_cur], g.data(), g.size() * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(e->cr.minstd3[e->rng_cur], m.data(), m.size() * 4, hipMemcpyHostToDevice));
    return CHUB_OK;
}

int chub_get_rng_compat_state(chub_env *e, uint32_t *state) {
    if (!e || !state) return fail(CHUB_ERR_ARG, "null argument");
    if (e->hp.rng_mode != CHUB_RNG_COMPAT) return fail(CHUB_ERR_ARG, "handle is not in COMPAT mode");
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipDeviceSynchronize());
    const size_t N = (size_t) e->hp.n_envs;
    std::vector<uint32_t> g, m;
    int rc;
    if ((rc = fetch(g, (const uint32_t *) e->cr.g3[e->rng_cur], N * 32)) || (rc = fetch(m, (const uint32_t *) e->cr.minstd3[e->rng_cur], N))) return rc;
    for (size_t i = 0; i < N; i++) {
        memcpy(&state[i * 33], &g[i * 32], 32 * 4);
        state[i * 33 + 32] = m[i];
    }
    return CHUB_OK;
}

int chub_profile_begin(chub_env *e, int max_steps, int every) {
    if (!e || max_steps <= 0 || every <= 0) return fail(CHUB_ERR_ARG, "bad argument");
    e->prof_every = every;
    e->prof_phase = 0;
    HIP_TRY(hipSetDevice(e->device));
    while (e->prof_events.size() < (size_t) max_steps * 4) {
        hipEvent_t ev;
        HIP_TRY(hipEventCreate(&ev));
        e->prof_events.push_back(ev);
    }
    e->prof_cap = (size_t) max_steps;
    e->prof_used = 0;
    e->prof_on = true;
    return CHUB_OK;
}

int chub_profile_end(chub_env *e, double *slot_ms_sum, double *env_ms_sum, int *n_steps) {
    if (!e || !slot_ms_sum || !env_ms_sum || !n_steps) return fail(CHUB_ERR_ARG, "null argument");
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipDeviceSynchronize());
    double a = 0, b = 0;
    for (size_t i = 0; i < e->prof_used; i++) {
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, e->prof_events[4 * i], e->prof_events[4 * i + 1]));
        a += ms;
        HIP_TRY(hipEventElapsedTime(&ms, e->prof_events[4 * i + 2], e->prof_events[4 * i + 3]));
        b += ms;
    }
    *slot_ms_sum = a;
    *env_ms_sum = b;
    *n_steps = (int) e->prof_used;
    e->prof_on = false;
    return CHUB_OK;
}

int chub_compat_replay_constructor(chub_env *e) {
    if (!e) return fail(CHUB_ERR_ARG, "null handle");
    e->walked_tick = 0;  // (a walk that ran ahead of its step is void)
    if (e->hp.rng_mode != CHUB_RNG_COMPAT) return fail(CHUB_ERR_ARG, "handle is not in COMPAT mode");
    HIP_TRY(hipSetDevice(e->device));
    int rc = sync_ctx(e, nullptr);
    if (rc) return rc;
    // (1) the two station constructors each run evs_reset (CHS.hpp:1152, 1462; AGG:188-196)
    StepArgs sa;
    memset(&sa, 0, sizeof sa);
    sa.station_filter = -1;
    sa.env_hi = (int32_t) (e->hp.n_envs - 1);
    sa.commit_rng = e->hp.compat_split;  // (the split form's walk leaves the streams' state in the shadow buffer: the commit moves rng_cur on)
    sa.rng_cur = e->rng_cur;
    launch_slot(true, e->hp, e->d_ctx, sa, nullptr, packed_ptrs(e), nullptr, nullptr);
    if (sa.commit_rng) e->rng_cur = (e->rng_cur + 1) % 3;
    e->empt_valid = e->hp.compat_split != 0;
    e->e2_tick = ~0u;  // (no pass has left empt2 for a walk two steps ahead: the first step counts for itself)
    // (2) HySystem.__init__: 101 hy_step()s with live FCEV arrivals (HYD:154-157,168,250-259) -> the streams advance and
    //     every env gets the hy_power_speed_list the reference would have built from its draws
    launch_compat_ctor_sweep(e->hp, e->d_ctx, e->rng_cur, nullptr);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    return CHUB_OK;
}

// ---- snapshot / restore (the reference cannot be pickled or deep-copied, MAIN:234) -----------
// Blob = header + the device arena (all state arrays; the tables in it are constant and simply ride along).
struct SnapshotHeader {
    uint64_t magic;
    int64_t n_envs, env_id0;
    chub_config cfg;
    int32_t rng_mode, t, price_count;
    uint32_t tick, graph_base;
    uint64_t arena_used;
    double hy_table[102];
    // per-env clocks (the clocks themselves are in the arena); the blob ends with every env's last tick
    int32_t predrawn, per_env;
    int32_t rng_cur, pad_;  // COMPAT: which of the arena's three stream buffers holds the committed streams
};
static const uint64_t kSnapMagic = 0x43485542534e4150ull;  // "CHUBSNAP"

int64_t chub_state_size(const chub_env *e) {
    if (!e) return fail(CHUB_ERR_ARG, "null handle");
    if (!e->arena) return fail(CHUB_ERR_UNSUPPORTED, "snapshot needs the single-arena allocation (CHUB_NO_ARENA is set)");
    return (int64_t) (sizeof(SnapshotHeader) + e->arena_used + (size_t) e->hp.n_envs * sizeof(uint32_t));
}

int chub_get_state(chub_env *e, void *buf, int64_t size) {
    if (!e || !buf) return fail(CHUB_ERR_ARG, "null argument");
    const int64_t need = chub_state_size(e);
    if (need < 0) return (int) need;
    if (size < need) return fail(CHUB_ERR_ARG, "buffer too small for the snapshot");
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipDeviceSynchronize());
    SnapshotHeader h;
    memset(&h, 0, sizeof h);
    h.magic = kSnapMagic;
    h.n_envs = e->hp.n_envs;
    h.env_id0 = e->hp.env_id0;
    h.cfg = e->cfg;
    h.rng_mode = e->hp.rng_mode;
    h.t = e->t;
    h.price_count = e->price_count;
    h.tick = e->tick;
    h.graph_base = e->graph_base;
    h.arena_used = e->arena_used;
    memcpy(h.hy_table, e->hy_table, sizeof h.hy_table);
    h.predrawn = e->predrawn ? 1 : 0;
    h.per_env = e->per_env ? 1 : 0;
    h.rng_cur = e->rng_cur;
    memcpy(buf, &h, sizeof h);
    HIP_TRY(hipMemcpy((char *) buf + sizeof h, e->arena, e->arena_used, hipMemcpyDeviceToHost));
    {
        const size_t N = (size_t) e->hp.n_envs;
        std::vector<int32_t> t(N);
        std::vector<uint32_t> tk(N);
        int rc = chub_env_clocks(e, t.data(), tk.data());
        if (rc) return rc;
        memcpy((char *) buf + sizeof h + e->arena_used, tk.data(), N * sizeof(uint32_t));
    }
    return CHUB_OK;
}

int chub_set_state(chub_env *e, const void *buf, int64_t size) {
    if (!e || !buf) return fail(CHUB_ERR_ARG, "null argument");
    e->walked_tick = 0;  // (a walk that ran ahead of its step is void)
    const int64_t need = chub_state_size(e);
    if (need < 0) return (int) need;
    SnapshotHeader h;
    if (size < (int64_t) sizeof h) return fail(CHUB_ERR_ARG, "snapshot truncated");
    memcpy(&h, buf, sizeof h);
    if (h.magic != kSnapMagic) return fail(CHUB_ERR_ARG, "not a chub snapshot");
    if (h.n_envs != e->hp.n_envs || h.env_id0 != e->hp.env_id0 || h.rng_mode != e->hp.rng_mode ||
        memcmp(&h.cfg, &e->cfg, sizeof h.cfg) != 0 || h.arena_used != e->arena_used || size < need)
        return fail(CHUB_ERR_ARG, "snapshot was taken from a handle with a different configuration");
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipDeviceSynchronize());
    // device pointers inside the arena are position-dependent: restore only into the handle's own layout, which the
    // checks above guarantee is the same; the DevCtx block (pointers, flags) is rewritten from the host copy
    HIP_TRY(hipMemcpy(e->arena, (const char *) buf + sizeof h, e->arena_used, hipMemcpyHostToDevice));
    e->empt_valid = false;  // (the restored slot state has not been counted)
    e->e2_tick = ~0u;       // (... nor has any pass left empt2 for a walk two steps ahead)
    if (h.rng_cur < 0 || h.rng_cur > 2) return fail(CHUB_ERR_ARG, "snapshot header is corrupt");
    e->rng_cur = h.rng_cur;
    e->t = h.t;
    e->price_count = h.price_count;
    e->tick = h.tick;
    e->graph_base = h.graph_base;
    memcpy(e->hy_table, h.hy_table, sizeof h.hy_table);
    e->ctx_dirty = true;
    {
        const size_t N = (size_t) e->hp.n_envs;
        e->predrawn = h.predrawn != 0;
        e->per_env = h.per_env != 0;
        e->h_tick.resize(N);
        memcpy(e->h_tick.data(), (const char *) buf + sizeof h + e->arena_used, N * sizeof(uint32_t));
        e->full_tick = 0;  // per env, from the blob
    }
    return CHUB_OK;
}

int chub_fcev_stuck_count(chub_env *e, int64_t *out) {
    if (!e || !out) return fail(CHUB_ERR_ARG, "null argument");
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipDeviceSynchronize());
    std::vector<uint8_t> f;
    int rc = fetch(f, (const uint8_t *) e->ev.hv_line, (size_t) e->hp.n_envs);
    if (rc) return rc;
    int64_t n = 0;
    for (uint8_t b : f) n += (b & 128u) ? 1 : 0;
    *out = n;
    return CHUB_OK;
}

int chub_set_ou_state(chub_env *e, const double *ou) {
    if (!e || !ou) return fail(CHUB_ERR_ARG, "null argument");
    HIP_TRY(hipSetDevice(e->device));
    const size_t N = (size_t) e->hp.n_envs;
    std::vector<double> t(3 * N);
    for (size_t i = 0; i < N; i++)
        for (int c = 0; c < 3; c++) t[(size_t) c * N + i] = ou[i * 3 + c];
    HIP_TRY(hipMemcpy(e->ev.ou, t.data(), t.size() * sizeof(double), hipMemcpyHostToDevice));
    return CHUB_OK;
}

int chub_get_hy_table(const chub_env *e, double *out102) {
    if (!e || !out102) return fail(CHUB_ERR_ARG, "null argument");
    memcpy(out102, e->hy_table, sizeof e->hy_table);
    return CHUB_OK;
}

int chub_get_hy_table_env(chub_env *e, int64_t env_index, double *out102) {
    if (!e || !out102) return fail(CHUB_ERR_ARG, "null argument");
    if (env_index < 0 || env_index >= e->hp.n_envs) return fail(CHUB_ERR_ARG, "env index out of range");
    if (!e->ev.hy_env) return chub_get_hy_table(e, out102);  // PHILOX: one table for the whole handle
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out102, (const double *) e->ev.hy_env + (size_t) env_index * 102, 102 * sizeof(double), hipMemcpyDeviceToHost));
    return CHUB_OK;
}

int chub_set_hy_table(chub_env *e, const double *in102) {
    if (!e || !in102) return fail(CHUB_ERR_ARG, "null argument");
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipDeviceSynchronize());
    memcpy(e->hy_table, in102, sizeof e->hy_table);
    HIP_TRY(hipMemcpy((void *) e->tb.hy_table, in102, sizeof e->hy_table, hipMemcpyHostToDevice));
    if (e->ev.hy_env) {
        const size_t N = (size_t) e->hp.n_envs;
        std::vector<double> rep(N * 102);
        for (size_t i = 0; i < N; i++) memcpy(&rep[i * 102], in102, sizeof e->hy_table);
        HIP_TRY(hipMemcpy(e->ev.hy_env, rep.data(), rep.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    return CHUB_OK;
}

}  // extern "C"
