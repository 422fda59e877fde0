// chub_kernels.hip -- the per-step hot path of the charging-hub environment as CDNA4 (gfx950) kernels.
//
//   k_slot_packed   2 charger slots per lane, every PHILOX step and reset (production).  The workgroup's 512 virtual lanes are
//           laid over whole envs end to end (hub-major 4-byte slot state, like the action rows); phases in the reference's
//           order (CHS.hpp:1188-1207 / 1499-1518): departures -> arrivals (pre-drawn levels, renege, balk) -> admission by
//           ballot + prefix rank (across waves through LDS) in the shadow of the class-row reads -> on/off, car_step = the
//           next row entry -> station sums (integer LDS atomics) -> last wave: add_car for the new cars, one 16-byte
//           station record per unit.
//   k_slot  the same phases with wave-local units (H = pow2 >= S_k lanes each): COMPAT streams (16-byte hot record, curves
//           evaluated), PHILOX scalar-load control.
//   k_env   lane = environment.  The scalar tail of step(): electrolyser clamp against the grid limit, FCEV arrivals +
//           SAE-J2601 fueling + the waiting list, electrolyser / compressor / tank, renewable netting, fuel cell, incomes and
//           reward, done, exogenous update (PV / wind / price OU) and the normalised observation.  The J2601 breakpoints
//           are immediates; the PV and wind rows of the current slot and the electrolyser action->power table are staged
//           in LDS; its last workgroups draw the next step's state-independent variates.  MULTI: per-env clocks.
//   k_reset_levels, k_draw_levels, k_compat_ctor_sweep, k_replay_soc, k_random_actions, k_tick_advance, k_fill_clocks: small
//           helpers (reset draws, a step's own station draws, COMPAT constructor replay, SoC introspection, bench policy,
//           graph replays, per-env clocks).
//
// No MFMA: there is no dense contraction anywhere in this path.  By bytes it is HBM-bound; measured VALU
// issue, the vector L1's access rate and HBM streaming are each more than half used.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

#include "chub_device.h"

namespace chub {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

// Per-env clocks: the envs a launch serves (StepArgs::env_mask) and the clock each env is on (StepArgs::env_clk, double-
// buffered by tick parity; lock-step: every env, the clock in StepArgs itself)
__device__ __forceinline__ bool in_group(const StepArgs &sa, int64_t env) { return !sa.env_mask || sa.env_mask[env] != 0; }
__device__ __forceinline__ uint32_t env_clk(const StepArgs &sa, int64_t n_envs, int64_t env) {
    return sa.env_clk[(int64_t) (sa.tick & 1u) * n_envs + env];
}
// A call on a subset of the envs only launches work for the range of envs it names (StepArgs::env_lo .. env_hi: the first and the
// last served env; every env in a lock-step call): lane i of a grid over `segments` copies of that range -> segment, env
__device__ __forceinline__ int64_t range_len(const StepArgs &sa) { return (int64_t) sa.env_hi - sa.env_lo + 1; }
__device__ __forceinline__ bool range_unit(const StepArgs &sa, int64_t i, int segments, int &seg, int64_t &env) {
    const int64_t R = range_len(sa);
    if (i >= (int64_t) segments * R) return false;
    seg = (i >= R ? 1 : 0) + (i >= 2 * R ? 1 : 0);
    env = sa.env_lo + (i - (int64_t) seg * R);
    return true;
}

__device__ __forceinline__ int clk_t(uint32_t c) { return (int) (c & 127u); }
__device__ __forceinline__ uint32_t clk_next(uint32_t c) {  // one step later: slot of day + 1 (mod 96), price_count + 1 (mod 4)
    return (uint32_t) ((clk_t(c) + 1) % 96) | ((((c >> 8) + 1u) & 3u) << 8);
}

// ------------------------------------------------------------------------------------------ Philox
struct U4 {
    uint32_t v[4];
};

__device__ __forceinline__ U4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                            uint32_t k1) {
#pragma unroll
    for (int i = 0; i < 10; i++) {
        // (one 32 x 32 -> 64 multiply per product -- v_mad_u64_u32 -- instead of a v_mul_hi_u32 / v_mul_lo_u32 pair: all three run at a quarter of the
        // 32-bit rate, and a block is 20 products)
        const uint64_t p0 = (uint64_t) 0xD2511F53u * c0, p1 = (uint64_t) 0xCD9E8D57u * c2;
        const uint32_t h0 = (uint32_t) (p0 >> 32), l0 = (uint32_t) p0, h1 = (uint32_t) (p1 >> 32), l1 = (uint32_t) p1;
        uint32_t n0 = h1 ^ c1 ^ k0, n2 = h0 ^ c3 ^ k1;
        c0 = n0; c1 = l1; c2 = n2; c3 = l0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    U4 r;
    r.v[0] = c0; r.v[1] = c1; r.v[2] = c2; r.v[3] = c3;
    return r;
}

__device__ __forceinline__ uint32_t pick(const U4 &b, int i) {
    uint32_t r = b.v[0];
    r = (i == 1) ? b.v[1] : r;
    r = (i == 2) ? b.v[2] : r;
    r = (i == 3) ? b.v[3] : r;
    return r;
}

// Philox tick of a launch = the host's tick (a kernel argument) + a base kept in device memory: a captured graph of steps bakes
// the host ticks in, and every replay moves the base on (k_tick_advance, the graph's last node), so no counter ever repeats
#define CHUB_TICK(hp, host_tick) ((host_tick) + *(hp).tick_base)
struct PhiloxCtx {
    uint32_t k0, k1, tick, gid;
    __device__ __forceinline__ U4 block(uint32_t site, uint32_t index, uint32_t blk) const {
        return philox4x32_10(blk, (site << 16) | index, tick, gid, k0, k1);
    }
};

// ------------------------------------------------------------- reference streams (COMPAT), one lane
// The ring lives in global memory (RingGlobal: every kernel but the split step's walk) or, for the walk of the split step, parked in LDS
// (RingLds: word w of the workgroup's lane l at s[w * 256 + l] -- a draw is then two LDS reads instead of two dependent global round trips)
struct RingGlobal {
    uint32_t *g;
    __device__ __forceinline__ uint32_t get(uint32_t i) const { return g[i]; }
    __device__ __forceinline__ void set(uint32_t i, uint32_t v) const { g[i] = v; }
};
template <int NW>
struct RingLdsT {  // NW: the lanes (envs) whose rings share the area
    uint32_t *s;
    __device__ __forceinline__ uint32_t get(uint32_t i) const { return s[i * (uint32_t) NW]; }
    __device__ __forceinline__ void set(uint32_t i, uint32_t v) const { s[i * (uint32_t) NW] = v; }
};
typedef RingLdsT<256> RingLds;
template <typename Ring>
struct CompatStreamT {
    Ring r;  // 32 words: ring[31] + front index
    uint32_t gf, gr, x;
    // (cur: which of the handle's three stream buffers holds the committed streams, StepArgs::rng_cur)
    __device__ void load(const CompatRng &cr, int cur, int64_t env) {  // (RingGlobal)
        r.g = cr.g3[cur] + env * 32;
        gf = r.get(31);
        gr = (gf + 28u) % 31u;
        x = cr.minstd3[cur][env];
    }
    __device__ void store(const CompatRng &cr, int cur, int64_t env) {
        r.set(31, gf);
        cr.minstd3[cur][env] = x;
    }
    // glibc random_r TYPE_3: *f += *r; result = *f >> 1  (rand(), CHS.hpp:41)
    __device__ uint32_t rand31() {
        uint32_t v = r.get(gf) + r.get(gr);
        r.set(gf, v);
        gf = (gf + 1u == 31u) ? 0u : gf + 1u;
        gr = (gr + 1u == 31u) ? 0u : gr + 1u;
        return v >> 1;
    }
    __device__ int level() { return (int) (rand31() % 1000u); }
    // std::minstd_rand0 (CHS.hpp:25): x * 16807 mod (2^31 - 1).  2^31 = 1 (mod 2^31 - 1), so the 46-bit product folds as low 31 bits +
    // the rest, minus the modulus once if need be: the same residue as the 64-bit modulo (x is never 0: the modulus is prime)
    __device__ uint32_t minstd() {
        const uint64_t p = (uint64_t) x * 16807ull;
        uint32_t v = (uint32_t) (p & 0x7FFFFFFFull) + (uint32_t) (p >> 31);
        x = v >= 2147483647u ? v - 2147483647u : v;
        return x;
    }
    // libstdc++ generate_canonical<double,53>: two draws.  The division by the constant R * R is q = sum * RN(1 / (R * R)), one exact
    // residual, one correction (chub_curves.h, CHUB_DIV_K64): with the correctly rounded reciprocal and q within an ulp of the quotient that is the
    // correctly rounded quotient (Markstein 1990) -- 3 instructions instead of the ~35 of the f64 division sequence, twice per polar trial
    // on the walk's serial chain (tests/test_host_cpu.py checks the identity on 4e8 dividends of this shape on the CPU)
    __device__ double canon_d() {
        const double R = 2147483646.0;
        double sum = (double) (minstd() - 1u);
        sum += (double) (minstd() - 1u) * R;
        double ret = CHUB_DIV_K64(sum, R * R);
        if (ret >= 1.0) ret = 0.99999999999999988898;
        return ret;
    }
    // generate_canonical<float,24>: one draw (the division by 2^31 is the multiplication by 2^-31: exact scaling, nothing underflows)
    __device__ float canon_f() {
        float ret = __fmul_rn((float) (minstd() - 1u), 4.656612873077392578125e-10f);
        if (ret >= 1.0f) ret = 0.99999994f;
        return ret;
    }
    // std::normal_distribution<double>, fresh per call: polar, y*mult (CHS.hpp:805).  In two halves: polar_d is all that touches the stream
    // (the rejection loop: cheap, serial per env); normal_of_polar_d is the transform (f64 log, division, sqrt: a hundred-odd instructions that
    // depend on nothing but the accepted point) -- the dense stream walk (compat_walk_wave) runs the first half one env per lane and the
    // second one CAR per lane.  normal_d = both, back to back: the same operations in the same order, the same bits.
    __device__ void polar_d(double &y_, double &r2) {
        double x_;
        do {
            x_ = 2.0 * canon_d() - 1.0;
            y_ = 2.0 * canon_d() - 1.0;
            r2 = x_ * x_ + y_ * y_;
        } while (r2 > 1.0 || r2 == 0.0);
    }
    static __device__ __forceinline__ double normal_of_polar_d(double y_, double r2, double mean, double sd) {
        double mult = __dsqrt_rn(-2.0 * log(r2) / r2);
        return (y_ * mult) * sd + mean;
    }
    __device__ double normal_d(double mean, double sd) {
        double y_, r2;
        polar_d(y_, r2);
        return normal_of_polar_d(y_, r2, mean, sd);
    }
    // std::normal_distribution<float> (CHS.hpp:819,833).  logf is evaluated as the f32 rounding of the f64 log.
    __device__ void polar_f(float &y_, float &r2) {
        float x_;
        do {
            x_ = __fsub_rn(__fmul_rn(2.0f, canon_f()), 1.0f);
            y_ = __fsub_rn(__fmul_rn(2.0f, canon_f()), 1.0f);
            r2 = __fadd_rn(__fmul_rn(x_, x_), __fmul_rn(y_, y_));
        } while (r2 > 1.0f || r2 == 0.0f);
    }
    static __device__ __forceinline__ float normal_of_polar_f(float y_, float r2, float mean, float sd) {
        float lg = (float) log((double) r2);
        float mult = __fsqrt_rn(__fdiv_rn(__fmul_rn(-2.0f, lg), r2));
        return __fadd_rn(__fmul_rn(__fmul_rn(y_, mult), sd), mean);
    }
    __device__ float normal_f(float mean, float sd) {
        float y_, r2;
        polar_f(y_, r2);
        return normal_of_polar_f(y_, r2, mean, sd);
    }
};
typedef CompatStreamT<RingGlobal> CompatStream;

// ------------------------------------------------------------------------------- small arithmetic
// RandomUtil::uniform_rand(a, b) at level k (CHS.hpp:35-44): two f32 roundings after the division
__device__ __forceinline__ float uniform_level(int k, float a, float b) {
    float tr = __fdiv_rn((float) k, 999.0f);
    return __fadd_rn(__fmul_rn(tr, __fsub_rn(b, a)), a);
}

// calculate_needed (CHS.hpp:879-898 / 1044-1063) from the cached curve times
__device__ __forceinline__ float emergency_of(float t_target, float t_soc, int tl) {
    float need = __fsub_rn(t_target, t_soc);
    if (need > 0.0f) {
        if ((float) tl <= ceilf(need)) return 10.0f;
        float q = __fdiv_rn(need, (float) tl);
        return (float) ((double) q * (double) q);
    }
    return 0.0f;
}

// Both uses the step makes of the emergency -- force-on when emergency >= 1.01 (CHS.hpp:1406) and min_power when
// emergency > 8 (CHS.hpp:1248) -- reduce to the "must charge" branch: otherwise emergency = (need/left)^2 with
// left > ceil(need) >= need, i.e. < 1.  So the hot path only needs this predicate, no division.
__device__ __forceinline__ bool must_charge(float t_target, float t_soc, int tl) {
    // need > 0 && tl <= ceil(need) (CHS.hpp:883-890): for an integer tl >= 1, ceil(need) >= tl exactly when need > tl - 1 (both
    // sides are exact in f32), and that implies need > 0.  tl <= 0 does not occur: an empty slot is never asked.
    const float need = __fsub_rn(t_target, t_soc);
    return need > (float) (tl - 1);
}

__device__ __forceinline__ float arrive_soc_from(double normal73) {  // mk_soc, CHS.hpp:804-814
    float d = (float) normal73;
    if ((double) d < 1.0) d = 1.0f;
    else if ((double) d > 10.0) d = 10.0f;
    return (float) (75.0 - 5.0 * (double) d);
}

// popcount of the bits of m below this lane: two mbcnt instructions (no lane mask to build, no 64-bit and)
__device__ __forceinline__ int prefix_count(uint64_t m) {
    return (int) __builtin_amdgcn_mbcnt_hi((uint32_t) (m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) m, 0u));
}

constexpr int MODE_COMPAT = 0, MODE_PHILOX = 1;

// XCD-aware work order.  The dispatcher hands workgroup b of a launch to XCD b % 8 (each XCD has its own L2); with this mapping XCD x works
// on ONE contiguous eighth of the launch's tiles (its b / 8-th) instead of on every eighth tile, so that neighbouring tiles -- which share
// cache lines at their borders -- share an L2 (slot kernel: L2 misses - 14 %, fabric writes - 11.5 %) and a CU touches one eighth of every
// array (vector-TLB misses 954 -> 0 per launch); nothing stays in L2 across kernel boundaries (profiles/round5_work_order_pmc.txt).  Measured (round 5, A/B inside one call,
// bit-identical; tiles, tail workgroups and level workgroups all in this order): cache-resident sizes gain 4-6 % of the step (C4 27.1 ->
// 25.95 us, 32 768 envs 18.3 -> 17.1, 8192 envs 10.8 -> 10.3, 131 072 envs 44.8 -> 43.8); the HBM-resident C5 LOSES 1 % (contiguous eighths
// concentrate each XCD's streams on fewer memory channels at a time), so handles of more than kXcdOrderSlots slots keep the dispatcher's
// order (HubParams::xcd -> PackedArgs::xcd / TailArgs::xcd).
__device__ __forceinline__ uint32_t xcd_order(uint32_t b, uint32_t nb, uint32_t on) {
    if (!on) return b;
    const uint32_t q = nb >> 3, r = nb & 7u, x = b & 7u, i = b >> 3;
    return x * q + (x < r ? x : r) + i;
}

// Phase timestamps of measurement builds (KFLAGS=-DCHUB_TRACE=1; nothing in a product build): the shader clock when the wave gets here
// -- a stamp waits for scalar results only, so a phase's share shows where the wave stood, not what it overlapped
#if CHUB_TRACE
#define CHUB_STAMP_DECL(n) unsigned long long stamp_[n] = {}
#define CHUB_STAMP(i) do { unsigned long long t_; asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); stamp_[i] = t_; } while (0)
// ... and the constant 100 MHz clock all XCDs share (every XCD has a shader-clock counter of its own): calibrates the stamps' rate
#define CHUB_STAMP_REAL(i) do { unsigned long long t_; asm volatile("s_memrealtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); stamp_[i] = t_; } while (0)
#else
#define CHUB_STAMP_REAL(i) do {} while (0)
#define CHUB_STAMP_DECL(n) do {} while (0)
#define CHUB_STAMP(i) do {} while (0)
#endif
// action_to_real (MGR:384-393) switches a pile on iff (a + 1) / 2 >= 0.5 on the f32 array.  In round-to-nearest-even f32
// that is exactly a >= -2^-25 (a + 1 rounds to 1 from -2^-25 upwards, the tie going to the even 1.0; checked against the
// two-step form on every f32 around the threshold and a stride over all others), so one compare replaces add, mul, compare.
constexpr float kActOnThreshold = -2.98023223876953125e-8f;

// What k_slot hands to the per-env tail for one (station, env) unit: one 16-byte record.
struct StationRec {
    float mn, chg, mx;  // min_power, charge_power, max_power (CHS.hpp:1257-1259)
    uint32_t pkd;       // pkd_make(line, flow_in, car_number), chub_device.h
};
__device__ __forceinline__ void rec_store(CHUB_G(uint32_t) rec, uint32_t u, float mn, float chg, float mx, uint32_t pkd) {
    u32x4 v = {__float_as_uint(mn), __float_as_uint(chg), __float_as_uint(mx), pkd};
    *((CHUB_G(u32x4)) (rec + 4u * u)) = v;
}
__device__ __forceinline__ StationRec rec_load(CHUB_G(uint32_t) rec, uint32_t u) {
    const u32x4 v = *((CHUB_G(u32x4)) (rec + 4u * u));
    StationRec r;
    r.mn = __uint_as_float(v.x); r.chg = __uint_as_float(v.y); r.mx = __uint_as_float(v.z); r.pkd = v.w;
    return r;
}

// DPP lane exchanges for the (integer) butterfly sums of the wave-local kernel.  After the xor-1 and xor-2 steps every lane of a quad holds the quad's
// sum, so the mirror patterns (lane i <-> 7-i, i <-> 15-i) pair the same partial sums as xor 4 / xor 8 would.
__device__ __forceinline__ int dppi_xor1(int v) { return __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true); }      // quad_perm [1,0,3,2]
__device__ __forceinline__ int dppi_xor2(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true); }      // quad_perm [2,3,0,1]
__device__ __forceinline__ int dppi_mirror8(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, true); }  // row_half_mirror
__device__ __forceinline__ int dppi_mirror16(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, true); } // row_mirror
// PHILOX-mode station sums are integer sums of slot powers in units of 2^-19 kW (see slot_body); back to kW in f32
__device__ __forceinline__ float fixed_to_kw(int v) { return (float) v * (1.0f / 524288.0f); }
// ... a slot power as that integer: the product with 2^19 is exact in f32, rounded to the nearest integer (ties to even), so a
// car's contribution is within 2^-20 kW of its f32 power and the error of a sum has no sign (round 3 truncated: up to 1.9e-6 kW
// per car, always downwards)
__device__ __forceinline__ int kw_to_fixed(float p) { return (int) rintf(p * 524288.0f); }

// PHILOX mode: mk_soc (CHS.hpp:804-814) from one 32-bit uniform by linear interpolation of the tabulated
// inverse CDF of clip(N(7,3),1,10): 12 bits pick the cell, 20 bits interpolate (three f32 roundings)
__device__ __forceinline__ float soc_from_word(const float *icdf, uint32_t w) {
    const uint32_t idx = w >> 20;
    const float frac = (float) (w & 0xFFFFFu) * (1.0f / 1048576.0f);
    const float a = icdf[idx], b = icdf[idx + 1];
    const float d = __fadd_rn(a, __fmul_rn(__fsub_rn(b, a), frac));
    return arrive_soc_from((double) d);
}
// PHILOX mode: mk_late_time (CHS.hpp:816-830) = #{j : w >= LT[j]} on the tabulated CDF of max(0, round(N(2,2)))
__device__ __forceinline__ int late_from_word(const uint32_t *thr, uint32_t w) {
    int late = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) late += (w >= thr[j]) ? 1 : 0;
    if (__any(w >= thr[7])) {  // the thresholds increase: beyond the 8th only with probability 0.3 %
#pragma unroll
        for (int j = 8; j < 16; j++) late += (w >= thr[j]) ? 1 : 0;
    }
    return late;
}

// PHILOX mode: N(0,1) from one 32-bit uniform by two-level tabulated inverse CDF (tools/gen_tables.py): 4096 cells,
// linear interpolation inside a cell; the lowest 16 cells (p < 2^-8: where the quantile function bends most) are refined by a
// second table with cells of 2^-20, the highest 16 are their mirror.
__device__ __forceinline__ float normal_from_word(const float *tz, const float *tl, uint32_t w) {
    const uint32_t cell = w >> 20;
    const bool hi = cell >= 4096u - 16u, lo = cell < 16u;
    const uint32_t m = hi ? ~w : w;  // tail: < 2^24
    const bool tail = hi || lo;
    const float *tab = tail ? tl : tz;
    const uint32_t idx = tail ? (m >> 12) : cell;
    const float frac = tail ? (float) (m & 0xFFFu) * (1.0f / 4096.0f) : (float) (w & 0xFFFFFu) * (1.0f / 1048576.0f);
    const float a = tab[idx], b = tab[idx + 1];
    const float z = __fadd_rn(a, __fmul_rn(__fsub_rn(b, a), frac));
    return hi ? -z : z;
}

// PHILOX: the station-level draws of receive_car (CHS.hpp:1272-1303 / 1583-1614) do not depend on state, only their
// use does: arrival count = table[slot of day][level]; queued car w stays iff level_w >= thr_renege[w]; arrival j stays
// iff level_j <= thr_balk[line + j], i.e. iff line <= inv_balk[level_j] - j (thr_balk is non-increasing).  So one lane
// per (station, env) unit draws them for the NEXT step, in extra blocks of the k_env launch.  Packed into 64 bits (the form
// chub_step_tape takes them in from the caller, pk_tape):
//   bits 0-9 renege pass bit per queue position, 10-13 arrivals n (<= 9), 14+4l (l = 0..10): how many of the n
//   arrivals stay (balk pass, incl. the j <= S guard) if the queue holds l cars after the renege pass.
// What those draws come to for a unit whose queue holds `line` cars (Station::line as the previous step left it): the renege
// pass over the queue (CHS.hpp:1286-1293), the arrivals, and -- slow station -- those of them that stay given the queue just
// thinned (CHS.hpp:1297-1306; the fast station records the un-thinned count, CHS.hpp:1617).  Decoded ONCE per unit, where the draws
// are made (one launch ahead), instead of by every lane of the unit in the slot kernel, which then only needs
//   assign = min(want, empties), line = min(want - assign, max_line)     (assign_car, CHS.hpp:417-430)
__device__ __forceinline__ uint32_t dk_make(uint64_t pk, int line, bool fast) {
    line = __popc((uint32_t) pk & ((1u << line) - 1u));
    const int n_in = (int) (pk >> 10) & 15;
    const int flow = fast ? n_in : (int) ((pk >> (14 + 4 * line)) & 15);
    return (uint32_t) (line + flow) | ((uint32_t) flow << 8);
}
// The same, drawn and decoded in one go where the queue length is known at draw time (the level blocks of k_env, k_draw_levels):
// only what this queue length needs -- `line` renege tests instead of all ten, the balk tests at ONE queue length instead of the
// whole table of eleven; the words: SITE_ARRIVE word 0 = arrival level, word 1 + j = balk level of arrival j; SITE_RENEGE word w = queue position w
__device__ __forceinline__ uint32_t draw_decoded_levels(const HubParams &hp, const Tables &tb, uint32_t tick, int t, int k, int64_t env,
                                                        int line, bool fast) {
    PhiloxCtx px{hp.key[0], hp.key[1], tick, (uint32_t) (hp.env_id0 + env)};
    const U4 a0 = px.block(SITE_ARRIVE, (uint32_t) k, 0);
    const U4 r0 = px.block(SITE_RENEGE, (uint32_t) k, 0);
    int n_in = (int) tb.cnt[k][t * kLevels + (int) (a0.v[0] % 1000u)];
    n_in = n_in > 9 ? 9 : n_in;
    uint32_t rw[10] = {r0.v[0], r0.v[1], r0.v[2], r0.v[3], 0u, 0u, 0u, 0u, 0u, 0u};
    if (__any(line > 4)) {  // the words of queue positions 4-9 only where some unit of the wave has such a queue
        const U4 r1 = px.block(SITE_RENEGE, (uint32_t) k, 1), r2 = px.block(SITE_RENEGE, (uint32_t) k, 2);
        rw[4] = r1.v[0]; rw[5] = r1.v[1]; rw[6] = r1.v[2]; rw[7] = r1.v[3]; rw[8] = r2.v[0]; rw[9] = r2.v[1];
    }
    int stay_q = 0;
#pragma unroll
    for (int w = 0; w < kMaxLine; w++) stay_q += (w < line && (int) (rw[w] % 1000u) >= (int) tb.thr_renege[w]) ? 1 : 0;
    int flow = n_in;
    if (!fast && __any(n_in > 0)) {  // the slow station records what survives the balk pass at the queue length just computed
        uint32_t bw[9] = {a0.v[1], a0.v[2], a0.v[3], 0u, 0u, 0u, 0u, 0u, 0u};
        if (__any(n_in > 3)) {
            const U4 a1 = px.block(SITE_ARRIVE, (uint32_t) k, 1), a2 = px.block(SITE_ARRIVE, (uint32_t) k, 2);
            bw[3] = a1.v[0]; bw[4] = a1.v[1]; bw[5] = a1.v[2]; bw[6] = a1.v[3]; bw[7] = a2.v[0]; bw[8] = a2.v[1];
        }
        const int S = hp.S[k];
        flow = 0;
#pragma unroll
        for (int j = 0; j < 9; j++) {
            if (j >= 3 && !__any(n_in > j)) break;
            const int top = (int) tb.inv_balk[bw[j] % 1000u] - j;  // arrival j stays iff line <= top (thr_balk is non-increasing)
            flow += (j < n_in && j <= S && stay_q <= top) ? 1 : 0;
        }
    }
    return (uint32_t) (stay_q + flow) | ((uint32_t) flow << 8);
}
__device__ __forceinline__ int dk_want(uint32_t dk) { return (int) (dk & 255u); }
__device__ __forceinline__ int dk_flow(uint32_t dk) { return (int) ((dk >> 8) & 255u); }

// ---------------------------------------------------------------------------------------- slot state
// COMPAT keeps, per slot, the 16-byte hot record of SlotArrays (power, t_target, t_soc, meta) and evaluates the charge
// curves in the step (the reference's streams make the arrival SoC a continuous value).
//
// PHILOX keeps FOUR bytes per slot and no curve is evaluated in the step at all.  The arrival SoC takes one of kSocLevels
// classes, and a car's whole charging history is the deterministic chain soc -> soc_to_time -> +1 slot -> time_to_soc /
// time_to_power (car_step, CHS.hpp:900-905 / 1065-1070) from that class: Tables::cls[k] holds, per class, (power, t_soc)
// after n = 0 .. kClsRow-1 car_steps, built once on the host with the same curve functions (chub_curves.h).  The slot keeps ONE word:
//   bits 0-4   stay_time - already_stay_time (0 = empty)        bit 5      charging this step
//   bits 6-10  n = car_steps taken since arrival                 bits 11-21 arrival-SoC class
//   bits 22-31 target-SoC level l (target = 80 + 20 * l / 999, CHS.hpp:35-44)
// and everything the step needs follows from two reads issued together once the word is there: one 16-byte row read (entries n
// and n + 1: where the car is on its curve and where one more car_step takes it) and soc_to_time(target) = Tables::ttab2[k][l], 4
// bytes of a 4 KB table that lives in every CU's vector L1.  Round 3 kept that f32 in a second state word (8 bytes per slot, 12 B
// read + 4 B written per slot and step); now a step reads 4 B and writes 4 B of state per slot.  stay_time itself (introspection
// only: Station::stay_time, CHS.hpp:245) goes to a cold byte array when the car is admitted.
__device__ __forceinline__ int ps_tl(uint32_t w) { return (int) (w & 31u); }
__device__ __forceinline__ uint32_t ps_n(uint32_t w) { return (w >> 6) & 31u; }
__device__ __forceinline__ uint32_t ps_cls(uint32_t w) { return (w >> 11) & 2047u; }
__device__ __forceinline__ uint32_t ps_lev(uint32_t w) { return w >> 22; }
__device__ __forceinline__ uint32_t ps_make(int stay, uint32_t cls, uint32_t lev) {  // a car that has just arrived
    return (uint32_t) stay | (cls << 11) | (lev << 22);
}
constexpr uint32_t kPsChg = 32u, kPsStep = 64u;  // the charging flag; one more car_step
static_assert(kSocLevels <= (1 << 11) && kLevels <= (1 << 10) && kClsRow <= 32, "field widths of the state word");
constexpr int kMaxStay = 31;  // 5-bit fields; chub_create refuses curves whose stays could exceed it

// car_step (CHS.hpp:900-905 / 1065-1070): soc and power one slot further along the curve, evaluated together.  Same
// expressions as time_to_soc / time_to_power (chub_curves.h), but the powers of x are shared between the two
// polynomials and each transcendental branch is skipped when no lane of the wave is in its regime.
template <int TYPE>
__device__ __forceinline__ void car_step_curves(float tt, bool cp, const CurveConsts &cc, float &soc, float &power) {
    if (cp) {
        soc = time_to_soc<TYPE>(tt, cp, cc);
        power = time_to_power<TYPE>(tt, cp);
        return;
    }
    if (TYPE == 0) {
        const float xf = tt * 15.0f;
        const double x = xf, t = tt;
        const bool p1 = tt >= 0 && t < (28.7 / 15), p2 = tt >= 0 && t < (51.2 / 15);  // fast_time_to_power
        const bool s0 = tt <= 0, s1 = t <= 28.7 / 15, s2 = t <= 51.2 / 15;            // fast_time_to_soc
        float pw = 0.0f, sc = 100.0f;
        if (__any(p1 || (!s0 && s1))) {
            const double e = exp(0.053 * x);
            if (p1) pw = (float) (0.7194 * e + 47.78);
            if (!s0 && s1) sc = (float) ((0.7194 / 0.053) * e + 50.15 * x - (double) cc.fast_aa1_c) * cc.fast_soc_scale;
        }
        if (__any((!p1 && p2) || (!s0 && !s1 && s2))) {
            const double x2 = x * x, x3 = x2 * x, x4 = x2 * x2, x5 = x4 * x;
            if (!p1 && p2) pw = (float) (0.0002253 * x4 - 0.03572 * x3 + 2.016 * x2 - 48.76 * x + 457.7);
            if (!s0 && !s1 && s2)
                sc = (float) ((0.0002253 / 5) * x5 - (0.03572 / 4) * x4 + (2.016 / 3) * x3 - (48.76 / 2) * x2 + 457.7 * x +
                              (double) cc.fast_aa2_c) * cc.fast_soc_scale;
        }
        if (s0) sc = 0.0f;
        soc = sc;
        power = pw;
    } else {
        const float xf = tt / 4.0f;
        const double x = xf, t = tt;
        const bool p1 = t < 2.33 * 4, p2 = t < 3.67 * 4;                 // slow_time_to_power
        const bool c0 = tt <= 0, c1 = t <= 2.33 * 4, c2 = t <= 3.67 * 4; // slow_c_hole
        float pw = 0.0f, ch = cc.slow_c2_end;
        if (__any(p1 || (!c0 && c1))) {
            const double x2 = x * x, x3 = x2 * x, x4 = x2 * x2, x5 = x4 * x;
            if (p1) pw = (float) (-0.002056 * x4 + 0.00921 * x3 + 0.03562 * x2 + 0.02379 * x + 6.007);
            if (!c0 && c1)
                ch = (float) (-0.0004112 * x5 + 0.0023025 * x4 + (0.03562 / 3) * x3 + 0.011895 * x2 + 6.007 * x);
        }
        if (__any((!p1 && p2) || (!c0 && !c1 && c2))) {
            if (!p1 && p2) pw = (float) ((-4.041 * x + 21.1) / (x - 0.485));
            if (!c0 && !c1 && c2) ch = (float) (-4.041 * x + 19.140115 * log(fabs(x - 0.485)) + 11.943306312699628);
        }
        if (c0) ch = 0.0f;
        soc = (float) CHUB_DIV_K64((double) (100.0f * ch), 19.285746346634653);
        power = pw;
    }
}

// COMPAT hot record, word y: the car's ARRIVAL SoC (round 6; until then soc_to_time(target), with the arrival SoC in a cold array of its own whose
// 4-byte store per new car was a read-modify-write of a whole sector).  soc_to_time(target) is one of 1000 values per station -- the target is
// level l of uniform_rand(80, 100) -- read from Tables::ttab[k] by the level kept in the record's word; chub_create checks once, on the device, that
// the table's entries are the bits the device's own soc_to_time gives (k_check_ttab): the table may then stand in for the function.
__device__ __forceinline__ int hot_level(uint32_t w) { return (int) ((w >> 15) & 1023u); }

// What add_car (CHS.hpp:864-877 / 1029-1042) produces for one admitted slot.
struct NewCar {
    float soc, t_target, t_soc, power;
    int stay, lev;  // lev: target-SoC level, target = 80 + 20 * lev / 999 (CHS.hpp:35-44)
};

template <int TYPE>
__device__ __forceinline__ NewCar make_car(float arrive_soc, int lev, float t_target, int late, bool cp) {
    NewCar c;
    c.soc = arrive_soc;
    c.lev = lev;
    c.t_target = t_target;
    c.t_soc = soc_to_time<TYPE>(arrive_soc, cp);
    const float need = __fsub_rn(c.t_target, c.t_soc);
    int stay = (int) ceilf(need) + late;  // calculate_min_charging_time + mk_late_time
    c.stay = stay > 127 ? 127 : stay;
    c.power = time_to_power<TYPE>(c.t_soc, cp);
    return c;
}

// ---- scalar-load control mode, evs_step(float) (CHS.hpp:1169-1186 / 1480-1497): one kW target per station; the piles are
// switched on in urgency order until the target is met (assign_on_off, CHS.hpp:1318-1362 / 1629-1674).  Wave-local units
// (H = pow2 >= S lanes).  Returns this lane's on / off decision.
template <int BLOCK>
__device__ __forceinline__ bool load_mode_on(const HubParams &hp, const StepArgs &sa, const StationArrays &st, const int k,
                                             const int env, const bool unit_ok, const bool valid, const int slot, const bool car0,
                                             const float pw0, const float em0, float *lds_f, uint32_t *lds_u, const int ubase) {
    const int wave = threadIdx.x >> 6;
    const int S = hp.S[k];
    const uint32_t sidx = (uint32_t) k * (uint32_t) hp.n_envs + (uint32_t) env;
    const bool cp = hp.constant_charging != 0;
    // catch_load (CHS.hpp:358-366): clamp to [min_power, max_power] of the previous calculate_output
    float load = 0.0f;
    if (unit_ok) {
        const StationRec pr = rec_load(st.rec, sidx);
        load = sa.actions[(uint32_t) env * (uint32_t) hp.act_dim + (uint32_t) (k ? hp.S[0] : 0)];
        if (load > pr.mx) load = pr.mx;
        else if (load < pr.mn) load = pr.mn;
    }
    // std::multimap keyed by -emergency (CHS.hpp:1324-1336): emergency descending, ties by slot index (ubase: the unit's first lane)
    int rk = 0;
    for (int j = 0; j < S; j++) {
        const float ej = __shfl(em0, ubase + j);
        rk += (ej > em0 || (ej == em0 && j < slot)) ? 1 : 0;
    }
    // rank_power_add (CHS.hpp:1375-1402): cumulative power of the cars in that order, added sequentially in f32
    float *by_rank = lds_f + wave * 64 + ubase;          // this unit's H floats
    uint32_t *car_by_rank = lds_u + wave * 64 + ubase;
    if (valid) {
        by_rank[rk] = car0 ? pw0 : 0.0f;
        car_by_rank[rk] = car0 ? 1u : 0u;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    float cum = 0.0f, mine = 0.0f;
    int cars_before = 0, mine_before = 0;
    for (int q = 0; q < S; q++) {
        if (q == rk) mine_before = cars_before;
        cum = __fadd_rn(cum, valid ? by_rank[q] : 0.0f);
        cars_before += valid ? (int) car_by_rank[q] : 0;
        if (q == rk) mine = cum;
    }
    bool chg;
    if (cp) {  // constant-power fleet: the first round(load / constant_power) cars in order
        const float constant_power = hp.type[k] == 0 ? (float) 36.44764034125146 : (float) 5.254973139368931;
        const int n_on = (int) roundf(__fdiv_rn(load, constant_power));
        chg = car0 && mine_before < n_on;
    } else {
        chg = car0 && ((double) load + 0.0001 >= (double) mine);
    }
    __builtin_amdgcn_wave_barrier();
    __syncthreads();  // the scratch areas are reused below
    return chg;
}

// ---------------------------------------------------------------------------------------- k_slot, COMPAT streams
// lane = charger slot; one (env, station) unit = H = pow2 >= S_k lanes, 64 / H units per wave.  Phases in the reference's order
// (CHS.hpp:1188-1207 / 1499-1518).  The unit's first lane walks the env's two reference streams in the reference's
// consumption order; station sums in the reference's sequential f32 order.  Pinned against the recorded reference
// trajectories (tests/golden).
// SPLIT (the split step of large batches): the walk was done by k_compat_walk, one ENV per lane (here one lane per UNIT walks while
// the other lanes of the wave idle); this body then only fetches what it came to -- flow, cars admitted, queue -- and the admitted
// lanes their car's variates.  Same draws in the same order, same arithmetic: bit-identical.
// wave0 / mid (k_compat_small): the station's units start at workgroup wave `wave0`, and mid() is called by every lane in front of the
// first look at what the walk came to (there: the workgroup barrier behind which the walker wave's results are in).
struct NoSlotMid {
    __device__ __forceinline__ void operator()() {}
};
template <int TYPE, bool RESET, int BLOCK, bool SPLIT = false, typename SlotMid = NoSlotMid>
__device__ __forceinline__ void slot_body_compat(const HubParams &hp, const StepArgs &sa, const SlotArrays &sl, const StationArrays &st,
                                 const CompatRng &cr, const Tables &tb, const int k, const int64_t block_local,
                                 float *lds_f, uint32_t *lds_u, const int wave0 = 0, SlotMid mid = SlotMid()) {
    constexpr int WAVES = BLOCK / 64;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave_abs = tid >> 6, wave = wave_abs - wave0;  // wave: among the station's waves (env mapping); wave_abs: the LDS areas
    // a unit takes U = S lanes (1 for a station without piles), floor(64 / U) units per wave, the rest of the wave idle: 3 units of 20
    // piles or 2 of 25 per wave (units of the next power of two left 38 % / 22 % of the lanes idle)
    const int U = hp.U[k], S = hp.S[k];
    const int upw = 64 / U;
    const int uiw = lane / U;
    const int slot = lane - uiw * U;
    const int64_t N = hp.n_envs;
    // 32-bit indices: n_envs * (piles + 2) < 2^31 is checked at create
    const int env_first = (int) block_local * (WAVES * upw);
    const int env = env_first + wave * upw + uiw;
    const bool unit_ok = uiw < upw && env < (int) N && in_group(sa, env);
    const bool valid = unit_ok && slot < S;
    const int leader = uiw < upw ? uiw * U : 63;  // the unit's first lane
    const uint64_t unit_mask = (U == 64) ? ~0ull : (uiw < upw ? (((1ull << U) - 1ull) << leader) : 0ull);
    const uint32_t idx = (uint32_t) hp.base[k] + (uint32_t) env * (uint32_t) S + (uint32_t) slot;
    const uint32_t sidx = (uint32_t) k * (uint32_t) N + (uint32_t) env;
    const int hub_slot = (k ? hp.S[0] : 0) + slot;
    const bool cp = hp.constant_charging != 0;

    // the unit's queue length, the slot's hot record and its action go out together (one memory round trip)
    uint32_t line_in = 0;
    u32x4 hot = {0u, 0u, 0u, 0u};
    float a = 0.0f;
    if (!RESET && unit_ok) line_in = st.rec[4u * sidx + 3u];
    if (!RESET && valid) {
        hot = ((CHUB_G(u32x4)) sl.hot)[idx];
        a = sa.actions[(uint32_t) env * (uint32_t) hp.act_dim + (uint32_t) hub_slot];
    }
    float power = __uint_as_float(hot.x), arr_soc = __uint_as_float(hot.y), t_soc = __uint_as_float(hot.z);
    int tl = (int) (hot.w & 127u);
    int meta = (int) (hot.w >> 8);  // all of the meta bits above the flag: stay_time | target level << 7 | car_steps << 17
    bool car = tl > 0, leave = false;
    float t_target = car ? tb.ttab[k][hot_level(hot.w)] : 0.0f;  // soc_to_time(target) of the car's level (hot_level above)
    int on_override = -1;
    if (!RESET && sa.load_mode)
        on_override = load_mode_on<BLOCK>(hp, sa, st, k, env, unit_ok, valid, slot, car, power,
                                          car ? emergency_of(t_target, t_soc, tl) : 0.0f, lds_f, lds_u, leader) ? 1 : 0;
    // judge_feasibility + assign_on_off_piece (CHS.hpp:1404-1413, 1364-1373); action_to_real (MGR:384-393)
    const bool on = on_override >= 0 ? (car && on_override != 0) : (car && (a >= kActOnThreshold || must_charge(t_target, t_soc, tl)));
    if (on) {  // car_step (CHS.hpp:900-905 / 1065-1070)
        meta += 1 << 17;  // one more car_step on this car's account (its SoC is replayed from it on demand)
        float soc_new;
        car_step_curves<TYPE>(__fadd_rn(t_soc, 1.0f), cp, hp.cc, soc_new, power);
        t_soc = soc_to_time<TYPE>(soc_new, cp);
    }
    if (car) {  // remove_car (CHS.hpp:912-923 / 1077-1088)
        tl -= 1;
        if (tl <= 0) {
            car = false;
            leave = true;
            tl = 0;
            power = t_target = t_soc = arr_soc = 0.0f;
            meta = 0;
        }
    }
    const bool charge = on && car;

    // ---- receive_car (CHS.hpp:1272-1316 / 1583-1627): arrivals, renege, balk, admission
    const bool empty = valid && !car;
    const uint64_t be = __ballot(empty) & unit_mask;
    const int empties = __popcll(be);
    const int rank = prefix_count(be);
    int line = pkd_line(line_in);
    const int mu = S / 2;  // round(charge_number / 2) on ints, CHS.hpp:1276
    // the unit's first lane walks the two reference streams in the reference's order and parks the per-admission
    // variates in LDS, indexed by admission rank
    float *lds_soc = lds_f;
    uint32_t *lds_lev = lds_u, *lds_late = lds_u + BLOCK;
    const int lbase = wave_abs * 64 + leader;
    int2 fa = make_int2(0, 0);
    int new_line = line;
    if (SPLIT) {
        mid();
        if (unit_ok && slot == 0) {
            const uint32_t w = st.fa[sa.tick & 1u][sidx];
            fa = make_int2((int) (int16_t) (w & 0xFFFFu), (int) ((w >> 16) & 255u));
            new_line = (int) (w >> 24);
        }
    } else if (unit_ok && slot == 0) {
        CompatStream rs;
        rs.load(cr, sa.rng_cur, env);
        int n_in;
        if (RESET) {
            const float cn = rs.normal_f((float) mu, 1.0f);
            int temp = (int) roundf(cn);
            temp = temp > mu + 3 ? mu + 3 : (temp < mu - 3 ? mu - 3 : temp);
            n_in = temp;
        } else {
            const int t_env = sa.env_clk ? clk_t(env_clk(sa, N, env)) : sa.t;  // per-env clocks: the env's own slot of day
            n_in = (int) tb.cnt[k][t_env * kLevels + rs.level()];
        }
        int tline = 0;
        for (int w = 0; w < line; w++) tline += (rs.level() >= (int) tb.thr_renege[w]) ? 1 : 0;
        new_line = tline;
        int true_in = 0;
        for (int j = 0; j < n_in; j++) {
            const int m = new_line + j;
            const int thr = (int) tb.thr_balk[m < kBalkTab ? m : kBalkTab - 1];
            true_in += (rs.level() <= thr && j <= S) ? 1 : 0;
        }
        const int fl = (TYPE == 0) ? n_in : true_in;
        int as = (new_line + fl) < empties ? (new_line + fl) : empties;
        new_line = new_line + fl - as;
        new_line = new_line < kMaxLine ? new_line : kMaxLine;
        for (int rr = 0; rr < as; rr++) {  // ascending slot order == ascending rank
            lds_soc[lbase + rr] = arrive_soc_from(rs.normal_d(7.0, 3.0));
            lds_lev[lbase + rr] = (uint32_t) rs.level();
            int late = (int) roundf(rs.normal_f(2.0f, 2.0f));  // mk_late_time("slow"), CHS.hpp:816-830
            lds_late[lbase + rr] = (uint32_t) (late < 0 ? 0 : late);
        }
        rs.store(cr, sa.rng_cur, env);
        fa = make_int2(fl, as);
    }
    const int flow = __shfl(fa.x, leader);
    const int assign = __shfl(fa.y, leader);
    line = __shfl(new_line, leader);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const bool adm = empty && rank < assign;
    float nc_soc = 0.0f;
    if (adm && SPLIT) {  // the car as the walk made it (compat_walk_env: add_car there, one record per admission rank)
        const uint32_t vi = (uint32_t) env * (uint32_t) (hp.S[0] + hp.S[1]) + (uint32_t) (k ? hp.S[0] : 0) + (uint32_t) rank;
        const u32x4 vv = ((CHUB_G(const u32x4)) sl.var[sa.tick & 1u])[2u * vi];
        nc_soc = __uint_as_float(sl.var[sa.tick & 1u][8u * vi + 4u]);
        power = __uint_as_float(vv.x);
        t_target = __uint_as_float(vv.y);
        t_soc = __uint_as_float(vv.z);
        tl = (int) (vv.w & 127u);
        car = tl > 0;
        meta = (int) vv.w;
    } else if (adm) {
        const int lev = (int) lds_lev[lbase + rank];
        const float target = uniform_level(lev, 80.0f, 100.0f);
        const NewCar nc = make_car<TYPE>(lds_soc[lbase + rank], lev, soc_to_time<TYPE>(target, cp), (int) lds_late[lbase + rank], cp);
        nc_soc = nc.soc;
        t_target = nc.t_target;
        t_soc = nc.t_soc;
        tl = nc.stay;
        power = nc.power;
        car = tl > 0;
        meta = nc.stay | (nc.lev << 7);
    } else if (leave || RESET) {
        meta = 0;
    }

    // ---- calculate_output (CHS.hpp:1233-1261 / 1544-1572): the reference adds slot powers sequentially in f32
    // (CHS.hpp:1244-1255): same order, same roundings
    const bool urgent = car && must_charge(t_target, t_soc, tl);
    const float v_min = urgent ? power : 0.0f, v_max = car ? power : 0.0f, v_chg = charge ? power : 0.0f;
    // every lane parks its three terms in the wave's own scratch area and the unit's first lane -- the only one that needs the sums --
    // adds them up in slot order (each lane fetching its S neighbours' terms by cross-lane reads cost 3 S of those per wave)
    float r_min = 0.0f, r_max = 0.0f, r_chg = 0.0f;
    {
        __builtin_amdgcn_wave_barrier();  // (the admitted lanes' reads of the walk's variates above are done)
        float *t_max = lds_f + wave_abs * 64, *t_min = (float *) lds_u + wave_abs * 64, *t_chg = (float *) lds_u + BLOCK + wave_abs * 64;
        t_max[lane] = v_max;
        t_min[lane] = v_min;
        t_chg[lane] = v_chg;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (unit_ok && slot == 0) {
            for (int i = 0; i < S; i++) {
                r_max = __fadd_rn(r_max, t_max[leader + i]);
                r_min = __fadd_rn(r_min, t_min[leader + i]);
                r_chg = __fadd_rn(r_chg, t_chg[leader + i]);
            }
        }
    }
    const int cars = __popcll(__ballot(car) & unit_mask);

    if (valid) {
        u32x4 h2;
        h2.x = __float_as_uint(power);
        h2.y = __float_as_uint(adm ? nc_soc : arr_soc);  // the arrival SoC (current SoC and target SoC are derived from it and the word on demand)
        h2.z = __float_as_uint(t_soc);
        h2.w = (uint32_t) tl | (charge ? 128u : 0u) | ((uint32_t) meta << 8);
        ((CHUB_G(u32x4)) sl.hot)[idx] = h2;
    }
    if (unit_ok && slot == 0) {
        rec_store(st.rec, sidx, r_min, r_chg, r_max, pkd_make(line, flow, cars));
    }
    if (SPLIT) {  // what the next step's walk needs of the slots: a slot is empty after remove_car iff it has at most one slot of stay left
        const int n_empty = __popcll(__ballot(valid && tl <= 1) & unit_mask), n_empty2 = __popcll(__ballot(valid && tl <= 2) & unit_mask);
        if (unit_ok && slot == 0) {
            st.empt[sidx] = (uint8_t) n_empty;
            st.empt2[sa.tick & 1u][sidx] = (uint8_t) n_empty2;
        }
        // (the step's draws are now taken: the buffer the walk left the streams' state in becomes the committed one -- on the host, by
        // moving StepArgs::rng_cur on: no copy)
    }
}

// ---------------------------------------------------------------------------------------- the split step's slot pass, two units' worth per lane
// slot_body_compat<.., SPLIT> spends its instructions on lanes that sit idle: the f64 curve work of car_step is needed by the third of
// the lanes whose car charges, add_car's by one lane in fifteen, and the sequential f32 sums by one lane per unit -- but a wave pays
// for each in full.  Here a lane carries TWO slots (virtual waves 2w and 2w + 1 of the same station), and those three pieces are done
// ONCE per wave for both: the charging lanes' inputs of both virtual waves are packed into the wave's own LDS area by ballot ranks (42 of
// 128 on average: one pass), evaluated by lanes 0 .. n - 1 with the same device functions -- the same bits -- and picked up again;
// likewise the new cars; and lane 2 * unit + v adds up the terms of unit `unit` of virtual wave v, so one loop serves both.  No
// workgroup barrier anywhere (the gathering across a WORKGROUP's waves lost to its barriers).  Lock-step and
// masked steps without the scalar-load mode; everything else keeps slot_body_compat.
template <int TYPE, int BLOCK>
__device__ __forceinline__ void slot_body_split2(const HubParams &hp, const StepArgs &sa, const SlotArrays &sl, const StationArrays &st,
                                                 const CompatRng &cr, const Tables &tb, const int k, const int64_t block_local, float *lds) {
    constexpr int WAVES = BLOCK / 64;
    constexpr int kArea = 3 * 128 + 16;  // per wave: three arrays of 128 terms (aliased by the gathering areas) + the units' words
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    // Units are laid end to end over the wave's 128 VIRTUAL lanes v = 64 j + lane (j = the lane's first / second slot): floor(128 / U) units per
    // wave -- 5 stations of 25 piles instead of the 2 x 2 that fit when a unit had to stay inside one virtual wave (round 6: a fifth of the
    // slow station's waves gone), 6 of 20 as before.  At most one unit straddles the two virtual waves; its ranks and counts take both ballots.
    const int U = hp.U[k], S = hp.S[k];
    const int upp = 128 / U;  // units per wave (U >= 8 here: at most 16)
    const uint32_t umagic = 65536u / (uint32_t) U + 1u;  // v / U = (v * umagic) >> 16, exact for v < 128 (the error term v / 65536 < 1 / U)
    const int N = (int) hp.n_envs;
    const int env_first = (int) block_local * (WAVES * upp);
    const uint32_t St = (uint32_t) (hp.S[0] + hp.S[1]);
    const bool cp = hp.constant_charging != 0;
    float *wl = lds + wave * kArea;
    uint32_t *wu = (uint32_t *) wl;

    int env[2], uiw[2], slot[2];
    bool unit_ok[2], valid[2];
    uint32_t idx[2], sidx[2], fa_w[2];
    uint64_t m_same[2], m_other[2];  // the unit's lanes in the lane's own virtual wave / in the other one
    int lead_lane[2], lead_j[2];     // where the unit's first virtual lane sits
    u32x4 hot[2];
    float a[2];
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const int v = 64 * j + lane;
        uiw[j] = (int) (((uint32_t) v * umagic) >> 16);
        slot[j] = v - uiw[j] * U;
        const int ua = uiw[j] * U, ub = ua + U;  // the unit's virtual lanes [ua, ub)
        const bool has = uiw[j] < upp;
        lead_lane[j] = has ? (ua & 63) : 63;
        lead_j[j] = ua >> 6;
        if (j == 0) {  // (ua < 64)
            const int hi = ub < 64 ? ub : 64;
            m_same[j] = has ? ((hi == 64 ? ~0ull : ((1ull << hi) - 1ull)) & ~((1ull << ua) - 1ull)) : 0ull;
            m_other[j] = (has && ub > 64) ? ((1ull << (ub - 64)) - 1ull) : 0ull;
        } else {       // (ub > 64)
            const int lo = ua > 64 ? ua - 64 : 0, hi = ub - 64 < 64 ? ub - 64 : 64;
            m_same[j] = has ? ((hi == 64 ? ~0ull : ((1ull << hi) - 1ull)) & ~((1ull << lo) - 1ull)) : 0ull;
            m_other[j] = (has && ua < 64) ? (~0ull << ua) : 0ull;
        }
        env[j] = env_first + wave * upp + uiw[j];
        unit_ok[j] = has && env[j] < N && in_group(sa, env[j]);
        valid[j] = unit_ok[j] && slot[j] < S;
        idx[j] = (uint32_t) hp.base[k] + (uint32_t) env[j] * (uint32_t) S + (uint32_t) slot[j];
        sidx[j] = (uint32_t) k * (uint32_t) N + (uint32_t) env[j];
        hot[j] = u32x4{0u, 0u, 0u, 0u};
        a[j] = 0.0f;
        fa_w[j] = 0u;
        if (unit_ok[j] && slot[j] == 0) fa_w[j] = st.fa[sa.tick & 1u][sidx[j]];  // what the walk came to (flow_in, cars admitted, queue): with the first loads
        if (valid[j]) {
            hot[j] = ((CHUB_G(u32x4)) sl.hot)[idx[j]];
            a[j] = sa.actions[(uint32_t) env[j] * (uint32_t) hp.act_dim + (uint32_t) ((k ? hp.S[0] : 0) + slot[j])];
        }
    }
    float power[2], t_target[2], t_soc[2], arr_soc[2];
    int tl[2], meta[2];
    bool car[2], leave[2], on[2], charge[2];
    CHUB_G(const float) ttab_k = tb.ttab[k];
#pragma unroll
    for (int j = 0; j < 2; j++) {
        power[j] = __uint_as_float(hot[j].x);
        arr_soc[j] = __uint_as_float(hot[j].y);
        t_soc[j] = __uint_as_float(hot[j].z);
        tl[j] = (int) (hot[j].w & 127u);
        meta[j] = (int) (hot[j].w >> 8);
        car[j] = tl[j] > 0;
        t_target[j] = car[j] ? ttab_k[hot_level(hot[j].w)] : 0.0f;  // soc_to_time(target) of the car's level (hot_level)
        leave[j] = false;
        // judge_feasibility + assign_on_off_piece (CHS.hpp:1404-1413, 1364-1373); action_to_real (MGR:384-393)
        on[j] = car[j] && (a[j] >= kActOnThreshold || must_charge(t_target[j], t_soc[j], tl[j]));
    }
    // ---- receive_car (CHS.hpp:1272-1316 / 1583-1627), ahead of car_step: who leaves depends on the stay alone (remove_car, CHS.hpp:912-923 /
    // 1077-1088), so the empties, the ranks and who is admitted are known here, and the new cars' variates (the walk's) are requested
    // BEFORE the curve work instead of behind it
    bool empty[2], adm[2];
    int rank[2], line[2], flow[2], assign[2];
    u32x4 vv[2];
    float nc_soc[2];
    uint64_t be[2];
#pragma unroll
    for (int j = 0; j < 2; j++) {
        empty[j] = valid[j] && !(car[j] && tl[j] > 1);
        be[j] = __ballot(empty[j]);
    }
#pragma unroll
    for (int j = 0; j < 2; j++) {
        // rank = the unit's empties in front of this lane: those of its own virtual wave below it, + (second virtual wave of a straddling
        // unit) all of the unit's empties in the first
        rank[j] = prefix_count(be[j] & m_same[j]) + (j == 1 ? __popcll(be[0] & m_other[1]) : 0);
        // the walk's word of the unit (flow_in, cars admitted, queue), from the unit's first virtual lane -- which for a lane's first slot is
        // in the first virtual wave, for its second slot in either
        const uint32_t w0 = (uint32_t) __shfl((int) fa_w[0], lead_lane[j]);
        uint32_t w = w0;
        if (j == 1) {
            const uint32_t w1 = (uint32_t) __shfl((int) fa_w[1], lead_lane[j]);
            w = lead_j[j] ? w1 : w0;
        }
        flow[j] = (int) (int16_t) (w & 0xFFFFu);
        assign[j] = (int) ((w >> 16) & 255u);
        line[j] = (int) (w >> 24);
        adm[j] = empty[j] && rank[j] < assign[j];
        vv[j] = u32x4{0u, 0u, 0u, 0u};
        nc_soc[j] = 0.0f;
        if (adm[j]) {  // the new car as the walk made it (add_car in compat_walk_env): requested here, picked up behind the curve work
            const uint32_t vi = (uint32_t) env[j] * St + (uint32_t) (k ? hp.S[0] : 0) + (uint32_t) rank[j];
            vv[j] = ((CHUB_G(const u32x4)) sl.var[sa.tick & 1u])[2u * vi];  // (the record's two 16-byte halves share a 32-byte sector)
            nc_soc[j] = __uint_as_float(sl.var[sa.tick & 1u][8u * vi + 4u]);
        }
    }
    // ---- car_step (CHS.hpp:900-905 / 1065-1070) of both virtual waves' charging cars, packed
    {
        const uint64_t b0 = __ballot(on[0]), b1 = __ballot(on[1]);
        const int n0 = __popcll(b0), n1 = __popcll(b1);
        if (n0 + n1 <= 64) {
            float *g_in = wl, *g_pw = wl + 64, *g_ts = wl + 128;
            const int p0 = prefix_count(b0), p1 = n0 + prefix_count(b1);
            if (on[0]) g_in[p0] = __fadd_rn(t_soc[0], 1.0f);
            if (on[1]) g_in[p1] = __fadd_rn(t_soc[1], 1.0f);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (lane < n0 + n1) {
                float soc_new, pw;
                car_step_curves<TYPE>(g_in[lane], cp, hp.cc, soc_new, pw);
                g_pw[lane] = pw;
                g_ts[lane] = soc_to_time<TYPE>(soc_new, cp);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (on[0]) { power[0] = g_pw[p0]; t_soc[0] = g_ts[p0]; }
            if (on[1]) { power[1] = g_pw[p1]; t_soc[1] = g_ts[p1]; }
            __builtin_amdgcn_wave_barrier();
        } else {  // (more than half of the slots charge: each virtual wave for itself)
#pragma unroll
            for (int j = 0; j < 2; j++)
                if (on[j]) {
                    float soc_new;
                    car_step_curves<TYPE>(__fadd_rn(t_soc[j], 1.0f), cp, hp.cc, soc_new, power[j]);
                    t_soc[j] = soc_to_time<TYPE>(soc_new, cp);
                }
        }
    }
#pragma unroll
    for (int j = 0; j < 2; j++) {
        if (on[j]) meta[j] += 1 << 17;  // one more car_step on this car's account (its SoC is replayed from it on demand)
        if (car[j]) {  // remove_car (CHS.hpp:912-923 / 1077-1088)
            tl[j] -= 1;
            if (tl[j] <= 0) {
                car[j] = false;
                leave[j] = true;
                tl[j] = 0;
                power[j] = t_target[j] = t_soc[j] = arr_soc[j] = 0.0f;
                meta[j] = 0;
            }
        }
        charge[j] = on[j] && car[j];
    }
    // ---- add_car (CHS.hpp:864-877 / 1029-1042): the walk evaluated it (one record per admission rank); the admitted lanes take theirs
#pragma unroll
    for (int j = 0; j < 2; j++) {
        if (adm[j]) {
            power[j] = __uint_as_float(vv[j].x);
            t_target[j] = __uint_as_float(vv[j].y);
            t_soc[j] = __uint_as_float(vv[j].z);
            tl[j] = (int) (vv[j].w & 127u);
            car[j] = tl[j] > 0;
            meta[j] = (int) vv[j].w;
        } else if (leave[j]) {
            meta[j] = 0;
        }
    }
    // ---- calculate_output (CHS.hpp:1233-1261 / 1544-1572): the reference adds slot powers sequentially in f32 (CHS.hpp:1244-1255):
    // same order, same roundings -- one lane per unit and term adds up the unit's S terms, which lie side by side in the 128 virtual lanes
    float *t_max = wl, *t_min = wl + 128, *t_chg = wl + 256;
    uint32_t *u_word = wu + 384;  // [unit]: line | flow | cars of the unit (pkd_make)
    {
        const uint64_t bc0 = __ballot(car[0]), bc1 = __ballot(car[1]);
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const bool urgent = car[j] && must_charge(t_target[j], t_soc[j], tl[j]);
            t_max[j * 64 + lane] = car[j] ? power[j] : 0.0f;
            t_min[j * 64 + lane] = urgent ? power[j] : 0.0f;
            t_chg[j * 64 + lane] = charge[j] ? power[j] : 0.0f;
            const int cars = __popcll((j ? bc1 : bc0) & m_same[j]) + __popcll((j ? bc0 : bc1) & m_other[j]);
            if (unit_ok[j] && slot[j] == 0) u_word[uiw[j]] = pkd_make(line[j], flow[j], cars);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // (one lane per unit AND term: lane 16 * term + unit -- a loop of S additions instead of 3 S; each lane stores its own word of the unit's
    // record: min, charge, max power in that order, the first of them the packed word behind)
    if (lane < 48 && (lane & 15) < upp) {
        const int term = lane >> 4, un = lane & 15;
        const int e = env_first + wave * upp + un;
        if (e < N && in_group(sa, e)) {
            const float *t = (term == 0 ? t_min : (term == 1 ? t_chg : t_max)) + un * U;
            float r = 0.0f;
            for (int i = 0; i < S; i++) r = __fadd_rn(r, t[i]);
            const uint32_t u = (uint32_t) k * (uint32_t) N + (uint32_t) e;
            st.rec[4u * u + (uint32_t) term] = __float_as_uint(r);
            if (term == 0) st.rec[4u * u + 3u] = u_word[un];
        }
    }
#pragma unroll
    for (int j = 0; j < 2; j++) {
        if (valid[j]) {
            u32x4 h2;
            h2.x = __float_as_uint(power[j]);
            h2.y = __float_as_uint(adm[j] ? nc_soc[j] : arr_soc[j]);  // the arrival SoC
            h2.z = __float_as_uint(t_soc[j]);
            h2.w = (uint32_t) tl[j] | (charge[j] ? 128u : 0u) | ((uint32_t) meta[j] << 8);
            // (the record goes back whole: storing only the words that changed -- 4 bytes for a car that neither charged, left nor arrived, nothing for
            // an empty slot -- was measured in round 6: the launch 49.6 -> 59 us, WRITE_SIZE 80 -> 100 MiB, FETCH_SIZE 54 -> 69 MiB: a partly
            // written sector costs a read-modify-write)
            ((CHUB_G(u32x4)) sl.hot)[idx[j]] = h2;
        }
    }
    {   // what the next step's walk needs of the slots: a slot is empty after remove_car iff it has at most one slot of stay left
        const uint64_t e1[2] = {__ballot(valid[0] && tl[0] <= 1), __ballot(valid[1] && tl[1] <= 1)};
        const uint64_t e2[2] = {__ballot(valid[0] && tl[0] <= 2), __ballot(valid[1] && tl[1] <= 2)};
#pragma unroll
        for (int j = 0; j < 2; j++) {
            if (unit_ok[j] && slot[j] == 0) {  // (the unit's first virtual lane: its own virtual wave's part + the other's)
                st.empt[sidx[j]] = (uint8_t) (__popcll(e1[j] & m_same[j]) + __popcll(e1[1 - j] & m_other[j]));
                st.empt2[sa.tick & 1u][sidx[j]] = (uint8_t) (__popcll(e2[j] & m_same[j]) + __popcll(e2[1 - j] & m_other[j]));
            }
        }
    }
}

// ---------------------------------------------------------------------------------------- k_slot, PHILOX, wave-local units
// The same phases on the 4-byte PHILOX slot state for what the packed kernel below does not cover: the scalar-load control mode
// (and every step of a handle created with chub_options.slot_kernel = 1: the parity cross-check).  One unit = H = pow2 >= S_k lanes of one wave: ballot + prefix rank inside the wave,
// integer DPP butterfly for the station sums.  Bit for bit the packed kernel's results (test_philox_other_slot_kernel runs
// every step through this one).
template <bool RESET, int BLOCK>
__device__ void slot_body_wave(const HubParams &hp, const StepArgs &sa, const SlotArrays &sl, const StationArrays &st,
                               const Tables &tb, const int k, const int64_t block_local, float *lds_f, uint32_t *lds_u) {
    constexpr int WAVES = BLOCK / 64;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int H = hp.H[k], S = hp.S[k], logH = hp.logH[k];
    const int upw = 64 >> logH;
    const int uiw = lane >> logH;
    const int slot = lane & (H - 1);
    const int64_t N = hp.n_envs;
    const int env_first = (int) block_local * (WAVES * upw);
    const int env = env_first + wave * upw + uiw;
    const bool unit_ok = env < (int) N && in_group(sa, env);
    const bool valid = unit_ok && slot < S;
    const uint64_t unit_mask = (H == 64) ? ~0ull : (((1ull << H) - 1ull) << (uiw << logH));
    const int hub_slot = (k ? hp.S[0] : 0) + slot;
    const uint32_t idx = (uint32_t) env * (uint32_t) (hp.S[0] + hp.S[1]) + (uint32_t) hub_slot;  // PHILOX state is hub-major
    const uint32_t sidx = (uint32_t) k * (uint32_t) N + (uint32_t) env;
    const bool fast = hp.type[k] == 0;
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    CHUB_G(const float) cls = tb.cls[k];

    uint32_t pk_in = 0;
    uint32_t w0 = 0u;
    float a = 0.0f;
    if (unit_ok) pk_in = st.pk[sa.tick & 1u][sidx];
    if (!RESET && valid) {
        w0 = sl.hot[idx];
        a = sa.actions[(uint32_t) env * (uint32_t) hp.act_dim + (uint32_t) hub_slot];
    }
    int tl = ps_tl(w0);
    bool car = tl > 0;
    f32x4 row = {0.0f, 0.0f, 0.0f, 0.0f};
    float t_target = 0.0f;
    if (car) {
        row = *(CHUB_G(const f32x4)) ((CHUB_G(const char)) cls + ((size_t) ps_cls(w0) * (kClsRow * 8u) + ps_n(w0) * 8u));
        t_target = tb.ttab[k][ps_lev(w0)];
    }
    float power = row.x, t_soc = row.y;
    int on_override = -1;
    if (!RESET && sa.load_mode)
        on_override = load_mode_on<BLOCK>(hp, sa, st, k, env, unit_ok, valid, slot, car, power,
                                          car ? emergency_of(t_target, t_soc, tl) : 0.0f, lds_f, lds_u, uiw << logH) ? 1 : 0;
    const bool on = on_override >= 0 ? (car && on_override != 0) : (car && (a >= kActOnThreshold || must_charge(t_target, t_soc, tl)));
    const bool step = on && tl > 1;  // a car that leaves this step is wiped right after its car_step (CHS.hpp:1196-1201)
    if (step) {  // car_step (CHS.hpp:900-905 / 1065-1070) = the next entry of the class row
        power = row.z;
        t_soc = row.w;
        w0 += kPsStep;
    }
    w0 &= ~kPsChg;
    if (car) {  // remove_car (CHS.hpp:912-923 / 1077-1088)
        tl -= 1;
        w0 -= 1u;
        if (tl <= 0) {
            car = false;
            w0 = 0u;
            power = t_target = t_soc = 0.0f;
        }
    }
    const bool charge = on && car;
    if (charge) w0 |= kPsChg;
    if (RESET) w0 = 0u;

    // ---- receive_car (CHS.hpp:1272-1316 / 1583-1627): arrivals, renege, balk, admission
    const bool empty = valid && !car;
    const uint64_t be = __ballot(empty) & unit_mask;
    const int empties = __popcll(be);
    const int rank = prefix_count(be);
    int line = 0, flow = 0, assign = 0;
    if (unit_ok) {
        int want;
        if (RESET) {
            // evs_reset: the unit's initial occupancy was drawn by k_reset_levels, one lane per unit, just before.  The fast
            // station records the raw draw, which is negative for small stations (mu - 3 < 0, CHS.hpp:1617, 832-842):
            // assign_car then admits nobody and the queue stays empty
            flow = fast ? (int) (int16_t) (pk_in & 0xFFFFu) : (int) ((pk_in >> 16) & 0xFFFFu);
            want = flow;
        } else {
            // this step's station-level draws were made and decoded one launch ahead (draw_decoded_levels)
            want = dk_want(pk_in);
            flow = dk_flow(pk_in);
        }
        assign = want < empties ? want : empties;  // assign_car, CHS.hpp:417-430
        line = want - assign;
        line = line < kMaxLine ? line : kMaxLine;
    }
    const bool adm = empty && rank < assign;
    if (adm) {  // add_car (CHS.hpp:864-877 / 1029-1042): one Philox block per new car, word 0 SoC class, 1 target level, 2 extra stay
        PhiloxCtx px{hp.key[0], hp.key[1], CHUB_TICK(hp, sa.tick), (uint32_t) (hp.env_id0 + env)};
        const U4 o = px.block(SITE_SOC, (uint32_t) hub_slot, 0);
        const uint32_t c = o.v[0] >> kSocLevelShift, lev = o.v[1] % 1000u;
        const f32x2 e0 = *(CHUB_G(const f32x2)) ((CHUB_G(const char)) cls + (size_t) c * (kClsRow * 8u));
        t_target = tb.ttab[k][lev];
        const int late = late_from_word(tb.late_thr, o.v[2]);
        int stay = (int) ceilf(__fsub_rn(t_target, e0.y)) + late;  // calculate_min_charging_time + mk_late_time
        stay = stay > kMaxStay ? kMaxStay : stay;
        power = e0.x;
        t_soc = e0.y;
        tl = stay;
        car = tl > 0;
        w0 = car ? ps_make(stay, c, lev) : 0u;
        sl.stay8[idx] = (uint8_t) stay;
    }

    // ---- calculate_output (CHS.hpp:1233-1261 / 1544-1572): order-independent sums -- every slot power rounded to the nearest
    // multiple of 2^-19 kW, integer butterfly over the unit's H lanes, one rounding to f32
    const bool urgent = car && must_charge(t_target, t_soc, tl);
    const int q = kw_to_fixed(power);
    int i_min = urgent ? q : 0, i_max = car ? q : 0, i_chg = charge ? q : 0;
    if (H > 1) { i_min += dppi_xor1(i_min); i_max += dppi_xor1(i_max); i_chg += dppi_xor1(i_chg); }
    if (H > 2) { i_min += dppi_xor2(i_min); i_max += dppi_xor2(i_max); i_chg += dppi_xor2(i_chg); }
    if (H > 4) { i_min += dppi_mirror8(i_min); i_max += dppi_mirror8(i_max); i_chg += dppi_mirror8(i_chg); }
    if (H > 8) { i_min += dppi_mirror16(i_min); i_max += dppi_mirror16(i_max); i_chg += dppi_mirror16(i_chg); }
    if (H > 16) { i_min += __shfl_xor(i_min, 16); i_max += __shfl_xor(i_max, 16); i_chg += __shfl_xor(i_chg, 16); }
    if (H > 32) { i_min += __shfl_xor(i_min, 32); i_max += __shfl_xor(i_max, 32); i_chg += __shfl_xor(i_chg, 32); }
    const int cars = __popcll(__ballot(car) & unit_mask);

    if (valid) sl.hot[idx] = w0;
    if (unit_ok && slot == 0) {
        rec_store(st.rec, sidx, fixed_to_kw(i_min), fixed_to_kw(i_chg), fixed_to_kw(i_max), pkd_make(line, flow, cars));
    }
}

// ---------------------------------------------------------------------------------------- k_slot_unit: one workgroup per unit
// Stations of more than 64 piles (the reference takes any size, CHS.hpp:1148, 1458) outside the packed production kernel: COMPAT
// streams, the scalar-load control mode, PHILOX handles forced onto the wave-local kernels.  One (env, station) unit = ONE
// workgroup of 256 lanes (S_k <= 256), lane = charger slot; the same phases as slot_body_compat / slot_body_wave with the
// wave-local steps made workgroup-wide through LDS: ballots of empty slots per wave, the stream walk / the decode of the
// pre-drawn levels on lane 0, the reference's sequential f32 sums (COMPAT) or 64-bit integer sums (PHILOX: what the packed
// kernel's BIG path adds up) by lane 0 over the slots in order.  A parity instrument, not a fast path.
template <bool RESET, int MODE>
__global__ __launch_bounds__(256) void k_slot_unit(const DevCtx *__restrict__ ctx, StepArgs sa, int k) {
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const HubParams &hp = ctx->hp;
    const SlotArrays &sl = ctx->sl;
    const StationArrays &st = ctx->st;
    const Tables &tb = ctx->tb;
    __shared__ float s_a[256], s_b[256], s_c[256];  // load mode: emergency, power, power by rank; sums: min, charge, max power per slot
    __shared__ uint32_t s_u[256], s_v[256], s_w[256];
    __shared__ float s_soc[256];                    // COMPAT: per-admission variates, by admission rank
    __shared__ uint64_t s_ball[8];
    __shared__ int s_hdr[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int env = (int) blockIdx.x;
    const int64_t N = hp.n_envs;
    if (!in_group(sa, env)) return;  // the whole workgroup: nothing of an env that is not served is touched
    const int S = hp.S[k], slot = tid;
    const bool valid = slot < S;
    const bool fast = hp.type[k] == 0, cp = hp.constant_charging != 0;
    const int hub_slot = (k ? hp.S[0] : 0) + slot;
    const uint32_t idx = MODE == MODE_PHILOX ? (uint32_t) env * (uint32_t) (hp.S[0] + hp.S[1]) + (uint32_t) hub_slot
                                             : (uint32_t) hp.base[k] + (uint32_t) env * (uint32_t) S + (uint32_t) slot;
    const uint32_t sidx = (uint32_t) k * (uint32_t) N + (uint32_t) env;
    CHUB_G(const float) cls = tb.cls[k];

    // ---- the slot as the previous step left it
    float power = 0.0f, t_target = 0.0f, t_soc = 0.0f, arr_soc = 0.0f, a = 0.0f, next_power = 0.0f, next_t_soc = 0.0f;
    int tl = 0, meta = 0;
    uint32_t w0 = 0u;
    if (!RESET && valid) {
        a = sa.actions[(uint32_t) env * (uint32_t) hp.act_dim + (uint32_t) hub_slot];
        if (MODE == MODE_COMPAT) {
            const u32x4 hot = ((CHUB_G(u32x4)) sl.hot)[idx];
            power = __uint_as_float(hot.x); arr_soc = __uint_as_float(hot.y); t_soc = __uint_as_float(hot.z);
            tl = (int) (hot.w & 127u);
            meta = (int) (hot.w >> 8);
            if (tl > 0) t_target = tb.ttab[k][hot_level(hot.w)];  // (hot_level: the record keeps the arrival SoC, the target's time is a table entry)
        } else {
            w0 = sl.hot[idx];
            tl = ps_tl(w0);
            if (tl > 0) {
                const f32x4 row = *(CHUB_G(const f32x4)) ((CHUB_G(const char)) cls + ((size_t) ps_cls(w0) * (kClsRow * 8u) + ps_n(w0) * 8u));
                power = row.x; t_soc = row.y; next_power = row.z; next_t_soc = row.w;
                t_target = tb.ttab[k][ps_lev(w0)];
            }
        }
    }
    bool car = tl > 0, leave = false;

    // ---- on / off: judge_feasibility + assign_on_off_piece (CHS.hpp:1404-1413, 1364-1373), or the scalar-load control
    // (assign_on_off, CHS.hpp:1318-1362 / 1629-1674: the piles in urgency order, std::multimap keyed by -emergency)
    bool on;
    if (!RESET && sa.load_mode) {
        const float em0 = car ? emergency_of(t_target, t_soc, tl) : 0.0f;
        s_a[tid] = em0;
        __syncthreads();
        int rk = 0;
        for (int j = 0; j < S; j++) {
            const float ej = s_a[j];
            rk += (ej > em0 || (ej == em0 && j < slot)) ? 1 : 0;
        }
        if (valid) {
            s_c[rk] = car ? power : 0.0f;
            s_u[rk] = car ? 1u : 0u;
        }
        __syncthreads();
        // catch_load (CHS.hpp:358-366) against the previous calculate_output; rank_power_add (CHS.hpp:1375-1402): sequential f32
        const StationRec pr = rec_load(st.rec, sidx);
        float load = sa.actions[(uint32_t) env * (uint32_t) hp.act_dim + (uint32_t) (k ? hp.S[0] : 0)];
        if (load > pr.mx) load = pr.mx;
        else if (load < pr.mn) load = pr.mn;
        float cum = 0.0f, mine = 0.0f;
        int cars_before = 0, mine_before = 0;
        for (int q = 0; q < S; q++) {
            if (q == rk) mine_before = cars_before;
            cum = __fadd_rn(cum, s_c[q]);
            cars_before += (int) s_u[q];
            if (q == rk) mine = cum;
        }
        if (cp) {
            const float constant_power = fast ? (float) 36.44764034125146 : (float) 5.254973139368931;
            const int n_on = (int) roundf(__fdiv_rn(load, constant_power));
            on = car && mine_before < n_on;
        } else {
            on = car && ((double) load + 0.0001 >= (double) mine);
        }
        __syncthreads();
    } else {
        on = car && (a >= kActOnThreshold || must_charge(t_target, t_soc, tl));
    }

    // ---- car_step (CHS.hpp:900-905 / 1065-1070), remove_car (CHS.hpp:912-923 / 1077-1088)
    if (MODE == MODE_COMPAT) {
        if (on) {
            meta += 1 << 17;
            float soc_new;
            const float tt = __fadd_rn(t_soc, 1.0f);
            if (fast) {
                car_step_curves<0>(tt, cp, hp.cc, soc_new, power);
                t_soc = soc_to_time<0>(soc_new, cp);
            } else {
                car_step_curves<1>(tt, cp, hp.cc, soc_new, power);
                t_soc = soc_to_time<1>(soc_new, cp);
            }
        }
        if (car) {
            tl -= 1;
            if (tl <= 0) {
                car = false; leave = true; tl = 0;
                power = t_target = t_soc = arr_soc = 0.0f;
                meta = 0;
            }
        }
    } else {
        if (on && tl > 1) {  // a car that leaves this step is wiped right after its car_step (CHS.hpp:1196-1201)
            power = next_power;
            t_soc = next_t_soc;
            w0 += kPsStep;
        }
        w0 &= ~kPsChg;
        if (car) {
            tl -= 1;
            w0 -= 1u;
            if (tl <= 0) {
                car = false;
                w0 = 0u;
                power = t_target = t_soc = 0.0f;
            }
        }
        if (RESET) w0 = 0u;
    }
    const bool charge = on && car;
    if (MODE == MODE_PHILOX && charge) w0 |= kPsChg;

    // ---- receive_car (CHS.hpp:1272-1316 / 1583-1627): empties over the whole unit, admission rank = empties below the slot
    const bool empty = valid && !car;
    const uint64_t be = __ballot(empty);
    if (lane == 0) s_ball[wave] = be;
    __syncthreads();
    int empties = 0, rank = prefix_count(be);
    for (int w = 0; w < 4; w++) {
        const int c = __popcll(s_ball[w]);
        empties += c;
        rank += w < wave ? c : 0;
    }
    if (tid == 0) {
        int line = (RESET || MODE == MODE_PHILOX) ? 0 : pkd_line(st.rec[4u * sidx + 3u]);
        int flow, assign;
        if (MODE == MODE_COMPAT) {
            const CompatRng &cr = ctx->cr;
            CompatStream rs;
            rs.load(cr, sa.rng_cur, env);
            const int mu = S / 2;  // round(charge_number / 2) on ints, CHS.hpp:1276
            int n_in;
            if (RESET) {
                int temp = (int) roundf(rs.normal_f((float) mu, 1.0f));
                n_in = temp > mu + 3 ? mu + 3 : (temp < mu - 3 ? mu - 3 : temp);
            } else {
                const int t_env = sa.env_clk ? clk_t(env_clk(sa, N, env)) : sa.t;
                n_in = (int) tb.cnt[k][t_env * kLevels + rs.level()];
            }
            int tline = 0;
            for (int w = 0; w < line; w++) tline += (rs.level() >= (int) tb.thr_renege[w]) ? 1 : 0;
            line = tline;
            int true_in = 0;
            for (int j = 0; j < n_in; j++) {
                const int m = line + j;
                const int thr = (int) tb.thr_balk[m < kBalkTab ? m : kBalkTab - 1];
                true_in += (rs.level() <= thr && j <= S) ? 1 : 0;
            }
            flow = fast ? n_in : true_in;  // CHS.hpp:1617 / 1306
            assign = (line + flow) < empties ? (line + flow) : empties;
            line = line + flow - assign;
            line = line < kMaxLine ? line : kMaxLine;
            for (int rr = 0; rr < assign; rr++) {  // ascending slot order == ascending rank
                s_soc[rr] = arrive_soc_from(rs.normal_d(7.0, 3.0));
                s_v[rr] = (uint32_t) rs.level();
                const int late = (int) roundf(rs.normal_f(2.0f, 2.0f));  // mk_late_time("slow"), CHS.hpp:816-830
                s_w[rr] = (uint32_t) (late < 0 ? 0 : late);
            }
            rs.store(cr, sa.rng_cur, env);
        } else {
            const uint32_t pk = st.pk[sa.tick & 1u][sidx];
            int want;
            if (RESET) {
                flow = fast ? (int) (int16_t) (pk & 0xFFFFu) : (int) ((pk >> 16) & 0xFFFFu);
                want = flow;
            } else {
                want = dk_want(pk);
                flow = dk_flow(pk);
            }
            assign = want < empties ? want : empties;
            line = want - assign;
            line = line < kMaxLine ? line : kMaxLine;
        }
        s_hdr[0] = assign; s_hdr[1] = flow; s_hdr[2] = line;
    }
    __syncthreads();
    const int assign = s_hdr[0];
    const bool adm = empty && rank < assign;
    float nc_soc = 0.0f;
    if (adm) {  // add_car (CHS.hpp:864-877 / 1029-1042)
        if (MODE == MODE_COMPAT) {
            const int lev = (int) s_v[rank];
            const float target = uniform_level(lev, 80.0f, 100.0f);
            const NewCar nc = fast ? make_car<0>(s_soc[rank], lev, soc_to_time<0>(target, cp), (int) s_w[rank], cp)
                                   : make_car<1>(s_soc[rank], lev, soc_to_time<1>(target, cp), (int) s_w[rank], cp);
            nc_soc = nc.soc;
            t_target = nc.t_target; t_soc = nc.t_soc; tl = nc.stay; power = nc.power;
            car = tl > 0;
            meta = nc.stay | (nc.lev << 7);
        } else {
            PhiloxCtx px{hp.key[0], hp.key[1], CHUB_TICK(hp, sa.tick), (uint32_t) (hp.env_id0 + env)};
            const U4 o = px.block(SITE_SOC, (uint32_t) hub_slot, 0);
            const uint32_t c = o.v[0] >> kSocLevelShift, lev = o.v[1] % 1000u;
            const f32x2 e0 = *(CHUB_G(const f32x2)) ((CHUB_G(const char)) cls + (size_t) c * (kClsRow * 8u));
            t_target = tb.ttab[k][lev];
            const int late = late_from_word(tb.late_thr, o.v[2]);
            int stay = (int) ceilf(__fsub_rn(t_target, e0.y)) + late;
            stay = stay > kMaxStay ? kMaxStay : stay;
            power = e0.x; t_soc = e0.y; tl = stay;
            car = tl > 0;
            w0 = car ? ps_make(stay, c, lev) : 0u;
            sl.stay8[idx] = (uint8_t) stay;
        }
    } else if (MODE == MODE_COMPAT && (leave || RESET)) {
        meta = 0;
    }

    // ---- calculate_output (CHS.hpp:1233-1261 / 1544-1572)
    const bool urgent = car && must_charge(t_target, t_soc, tl);
    const uint64_t bc = __ballot(car);
    if (lane == 0) s_ball[4 + wave] = bc;
    if (MODE == MODE_COMPAT) {
        s_a[tid] = urgent ? power : 0.0f;
        s_b[tid] = charge ? power : 0.0f;
        s_c[tid] = car ? power : 0.0f;
    } else {
        const int q = kw_to_fixed(power);
        s_u[tid] = (uint32_t) (urgent ? q : 0);
        s_v[tid] = (uint32_t) (charge ? q : 0);
        s_w[tid] = (uint32_t) (car ? q : 0);
    }
    if (valid) {
        if (MODE == MODE_COMPAT) {
            u32x4 h2;
            h2.x = __float_as_uint(power); h2.y = __float_as_uint(adm ? nc_soc : arr_soc); h2.z = __float_as_uint(t_soc);
            h2.w = (uint32_t) tl | (charge ? 128u : 0u) | ((uint32_t) meta << 8);
            ((CHUB_G(u32x4)) sl.hot)[idx] = h2;
        } else {
            sl.hot[idx] = w0;
        }
    }
    __syncthreads();
    if (tid == 0) {
        const int cars = __popcll(s_ball[4]) + __popcll(s_ball[5]) + __popcll(s_ball[6]) + __popcll(s_ball[7]);
        float r_min, r_chg, r_max;
        if (MODE == MODE_COMPAT) {  // the reference adds the slot powers sequentially in f32 (CHS.hpp:1244-1255)
            r_min = r_chg = r_max = 0.0f;
            for (int i = 0; i < S; i++) {
                r_max = __fadd_rn(r_max, s_c[i]);
                r_min = __fadd_rn(r_min, s_a[i]);
                r_chg = __fadd_rn(r_chg, s_b[i]);
            }
        } else {  // order-independent: 64-bit sums of the slot powers in units of 2^-19 kW, one rounding to f32
            long long i_min = 0, i_chg = 0, i_max = 0;
            for (int i = 0; i < S; i++) {
                i_min += (long long) (int) s_u[i];
                i_chg += (long long) (int) s_v[i];
                i_max += (long long) (int) s_w[i];
            }
            r_min = (float) i_min * (1.0f / 524288.0f);
            r_chg = (float) i_chg * (1.0f / 524288.0f);
            r_max = (float) i_max * (1.0f / 524288.0f);
        }
        rec_store(st.rec, sidx, r_min, r_chg, r_max, pkd_make(s_hdr[2], s_hdr[1], cars));
    }
}

// ---------------------------------------------------------------------------------------- k_slot_unit_any: a unit of any size
// Stations of more than 256 piles (the reference's constructors take any count, CHS.hpp:1148, 1458).  One (env, station) unit = ONE
// workgroup of 256 lanes that goes over the unit's piles in chunks of 256, in slot order, in two passes with the slot records
// themselves (global memory; every lane re-reads only what it wrote) as the state between them:
//   pass 1  per chunk: on / off, car_step, departures, the record written back; the unit's empty slots counted
//   lane 0  arrivals / renege / balk / how many are admitted (receive_car; COMPAT: the env's streams in the reference's order)
//   pass 2  per chunk: admission by rank (COMPAT: lane 0 draws the chunk's new cars' variates, in rank order, as it gets there),
//           add_car, then calculate_output: lane 0 adds the chunk's slot powers to the running sums IN SLOT ORDER (COMPAT: the
//           reference's sequential f32 sums; PHILOX: 64-bit integers)
// The scalar-load control (evs_step(float)) ranks every pile of the unit by urgency first: three arrays over the whole unit in
// LDS, hence kMaxPiles.  A parity instrument like k_slot_unit (same phases, same arithmetic), not a fast path.
template <bool RESET, int MODE>
__global__ __launch_bounds__(256) void k_slot_unit_any(const DevCtx *__restrict__ ctx, StepArgs sa, int k) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const HubParams &hp = ctx->hp;
    const SlotArrays &sl = ctx->sl;
    const StationArrays &st = ctx->st;
    const Tables &tb = ctx->tb;
    __shared__ float s_a[256], s_b[256], s_c[256];  // the chunk's min / charge / max power per slot (COMPAT)
    __shared__ uint32_t s_u[256], s_v[256], s_w[256];  // ... as integers (PHILOX); COMPAT: s_v / s_w = level, late time by admission rank
    __shared__ float s_soc[256];                    // COMPAT: arrival SoC by admission rank within the chunk
    __shared__ uint64_t s_ball[8];
    __shared__ int s_hdr[4];
    __shared__ int s_cnt;
    __shared__ float s_em[kMaxPiles], s_pw[kMaxPiles];    // load mode: emergency by slot; power by urgency rank -> running sum
    __shared__ uint16_t s_cb[kMaxPiles], s_rk[kMaxPiles]; // ... cars by urgency rank -> cars before the rank; urgency rank by slot
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int env = (int) blockIdx.x;
    const int64_t N = hp.n_envs;
    if (!in_group(sa, env)) return;  // the whole workgroup: nothing of an env that is not served is touched
    const int S = hp.S[k], chunks = (S + 255) >> 8;
    const bool fast = hp.type[k] == 0, cp = hp.constant_charging != 0;
    const int hub0 = k ? hp.S[0] : 0;
    const uint32_t idx0 = MODE == MODE_PHILOX ? (uint32_t) env * (uint32_t) (hp.S[0] + hp.S[1]) + (uint32_t) hub0
                                              : (uint32_t) hp.base[k] + (uint32_t) env * (uint32_t) S;
    const uint32_t sidx = (uint32_t) k * (uint32_t) N + (uint32_t) env;
    CHUB_G(const float) cls = tb.cls[k];
    if (tid == 0) s_cnt = 0;
    __syncthreads();

    // ---- scalar-load control (assign_on_off, CHS.hpp:1318-1362 / 1629-1674): the piles in urgency order (std::multimap keyed by
    // -emergency: ties in slot order), switched on until the target is met
    float load = 0.0f;
    if (!RESET && sa.load_mode) {
        for (int slot = tid; slot < S; slot += 256) {
            const uint32_t idx = idx0 + (uint32_t) slot;
            float t_target = 0.0f, t_soc = 0.0f;
            int tl;
            if (MODE == MODE_COMPAT) {
                const u32x4 hot = ((CHUB_G(u32x4)) sl.hot)[idx];
                t_soc = __uint_as_float(hot.z);
                tl = (int) (hot.w & 127u);
                if (tl > 0) t_target = tb.ttab[k][hot_level(hot.w)];
            } else {
                const uint32_t w0 = sl.hot[idx];
                tl = ps_tl(w0);
                if (tl > 0) {
                    t_soc = (*(CHUB_G(const f32x2)) ((CHUB_G(const char)) cls + ((size_t) ps_cls(w0) * (kClsRow * 8u) + ps_n(w0) * 8u))).y;
                    t_target = tb.ttab[k][ps_lev(w0)];
                }
            }
            s_em[slot] = tl > 0 ? emergency_of(t_target, t_soc, tl) : 0.0f;
        }
        __syncthreads();
        for (int slot = tid; slot < S; slot += 256) {
            const uint32_t idx = idx0 + (uint32_t) slot;
            float power = 0.0f;
            bool car;
            if (MODE == MODE_COMPAT) {
                const u32x4 hot = ((CHUB_G(u32x4)) sl.hot)[idx];
                car = (hot.w & 127u) != 0u;
                power = __uint_as_float(hot.x);
            } else {
                const uint32_t w0 = sl.hot[idx];
                car = ps_tl(w0) > 0;
                if (car) power = (*(CHUB_G(const f32x2)) ((CHUB_G(const char)) cls + ((size_t) ps_cls(w0) * (kClsRow * 8u) + ps_n(w0) * 8u))).x;
            }
            const float em0 = s_em[slot];
            int rk = 0;
            for (int j = 0; j < S; j++) {
                const float ej = s_em[j];
                rk += (ej > em0 || (ej == em0 && j < slot)) ? 1 : 0;
            }
            s_rk[slot] = (uint16_t) rk;
            s_pw[rk] = car ? power : 0.0f;
            s_cb[rk] = car ? 1 : 0;
        }
        __syncthreads();
        // catch_load (CHS.hpp:358-366) against the previous calculate_output; rank_power_add (CHS.hpp:1375-1402): sequential f32
        const StationRec pr = rec_load(st.rec, sidx);
        load = sa.actions[(uint32_t) env * (uint32_t) hp.act_dim + (uint32_t) hub0];
        if (load > pr.mx) load = pr.mx;
        else if (load < pr.mn) load = pr.mn;
        if (tid == 0) {
            float cum = 0.0f;
            int cars_before = 0;
            for (int q = 0; q < S; q++) {
                const int c = (int) s_cb[q];
                cum = __fadd_rn(cum, s_pw[q]);
                s_pw[q] = cum;                        // the running sum up to and including rank q
                s_cb[q] = (uint16_t) cars_before;     // cars of a higher urgency
                cars_before += c;
            }
        }
        __syncthreads();
    }

    // ---- pass 1: the slots as the previous step left them -> on / off, car_step, departures
    for (int c = 0; c < chunks; c++) {
        const int slot = (c << 8) + tid;
        const bool valid = slot < S;
        const uint32_t idx = idx0 + (uint32_t) slot;
        float power = 0.0f, t_target = 0.0f, t_soc = 0.0f, arr_soc = 0.0f, a = 0.0f;
        int tl = 0, meta = 0;
        uint32_t w0 = 0u;
        if (!RESET && valid) {
            a = sa.actions[(uint32_t) env * (uint32_t) hp.act_dim + (uint32_t) (hub0 + slot)];
            if (MODE == MODE_COMPAT) {
                const u32x4 hot = ((CHUB_G(u32x4)) sl.hot)[idx];
                power = __uint_as_float(hot.x); arr_soc = __uint_as_float(hot.y); t_soc = __uint_as_float(hot.z);
                tl = (int) (hot.w & 127u);
                meta = (int) (hot.w >> 8);
                if (tl > 0) t_target = tb.ttab[k][hot_level(hot.w)];
            } else {
                w0 = sl.hot[idx];
                tl = ps_tl(w0);
                if (tl > 0) {  // (the state word says the rest: a car_step is one more entry along the class row)
                    t_soc = (*(CHUB_G(const f32x2)) ((CHUB_G(const char)) cls + ((size_t) ps_cls(w0) * (kClsRow * 8u) + ps_n(w0) * 8u))).y;
                    t_target = tb.ttab[k][ps_lev(w0)];
                }
            }
        }
        bool car = tl > 0;
        bool on;
        if (!RESET && sa.load_mode) {
            on = false;
            if (car) {
                const int rk = (int) s_rk[slot];
                if (cp) {
                    const float constant_power = fast ? (float) 36.44764034125146 : (float) 5.254973139368931;
                    const int n_on = (int) roundf(__fdiv_rn(load, constant_power));
                    on = (int) s_cb[rk] < n_on;
                } else {
                    on = (double) load + 0.0001 >= (double) s_pw[rk];
                }
            }
        } else {
            on = car && (a >= kActOnThreshold || must_charge(t_target, t_soc, tl));
        }
        // car_step (CHS.hpp:900-905 / 1065-1070), remove_car (CHS.hpp:912-923 / 1077-1088)
        if (MODE == MODE_COMPAT) {
            if (on) {
                meta += 1 << 17;
                float soc_new;
                const float tt = __fadd_rn(t_soc, 1.0f);
                if (fast) {
                    car_step_curves<0>(tt, cp, hp.cc, soc_new, power);
                    t_soc = soc_to_time<0>(soc_new, cp);
                } else {
                    car_step_curves<1>(tt, cp, hp.cc, soc_new, power);
                    t_soc = soc_to_time<1>(soc_new, cp);
                }
            }
            if (car) {
                tl -= 1;
                if (tl <= 0) {
                    car = false; tl = 0;
                    power = t_soc = arr_soc = 0.0f;
                    meta = 0;
                }
            }
            if (RESET) meta = 0;
        } else {
            if (on && tl > 1) w0 += kPsStep;  // a car that leaves this step is wiped right after its car_step (CHS.hpp:1196-1201)
            w0 &= ~kPsChg;
            if (car) {
                tl -= 1;
                w0 -= 1u;
                if (tl <= 0) { car = false; w0 = 0u; }
            }
            if (RESET) w0 = 0u;
        }
        const bool charge = on && car;
        if (valid) {
            if (MODE == MODE_COMPAT) {
                u32x4 h2;
                h2.x = __float_as_uint(power); h2.y = __float_as_uint(arr_soc); h2.z = __float_as_uint(t_soc);
                h2.w = (uint32_t) tl | (charge ? 128u : 0u) | ((uint32_t) meta << 8);
                ((CHUB_G(u32x4)) sl.hot)[idx] = h2;
            } else {
                sl.hot[idx] = charge ? (w0 | kPsChg) : w0;
            }
        }
        const uint64_t be = __ballot(valid && !car);
        if (lane == 0 && be) atomicAdd(&s_cnt, __popcll(be));
    }
    __syncthreads();
    const int empties = s_cnt;

    // ---- receive_car (CHS.hpp:1272-1316 / 1583-1627): lane 0 decides how many cars the unit admits
    CompatStream rs;
    if (tid == 0) {
        int line = (RESET || MODE == MODE_PHILOX) ? 0 : pkd_line(st.rec[4u * sidx + 3u]);
        int flow, assign;
        if (MODE == MODE_COMPAT) {
            rs.load(ctx->cr, sa.rng_cur, env);
            const int mu = S / 2;  // round(charge_number / 2) on ints, CHS.hpp:1276
            int n_in;
            if (RESET) {
                int temp = (int) roundf(rs.normal_f((float) mu, 1.0f));
                n_in = temp > mu + 3 ? mu + 3 : (temp < mu - 3 ? mu - 3 : temp);
            } else {
                const int t_env = sa.env_clk ? clk_t(env_clk(sa, N, env)) : sa.t;
                n_in = (int) tb.cnt[k][t_env * kLevels + rs.level()];
            }
            int tline = 0;
            for (int w = 0; w < line; w++) tline += (rs.level() >= (int) tb.thr_renege[w]) ? 1 : 0;
            line = tline;
            int true_in = 0;
            for (int j = 0; j < n_in; j++) {
                const int m = line + j;
                const int thr = (int) tb.thr_balk[m < kBalkTab ? m : kBalkTab - 1];
                true_in += (rs.level() <= thr && j <= S) ? 1 : 0;
            }
            flow = fast ? n_in : true_in;  // CHS.hpp:1617 / 1306
            assign = (line + flow) < empties ? (line + flow) : empties;
            line = line + flow - assign;
            line = line < kMaxLine ? line : kMaxLine;
        } else {
            const uint32_t pk = st.pk[sa.tick & 1u][sidx];
            int want;
            if (RESET) {
                flow = fast ? (int) (int16_t) (pk & 0xFFFFu) : (int) ((pk >> 16) & 0xFFFFu);
                want = flow;
            } else {
                want = dk_want(pk);
                flow = dk_flow(pk);
            }
            assign = want < empties ? want : empties;
            line = want - assign;
            line = line < kMaxLine ? line : kMaxLine;
        }
        s_hdr[0] = assign; s_hdr[1] = flow; s_hdr[2] = line;
    }
    __syncthreads();
    const int assign = s_hdr[0];

    // ---- pass 2: admission by rank, add_car (CHS.hpp:864-877 / 1029-1042), calculate_output (CHS.hpp:1233-1261 / 1544-1572)
    int base = 0, cars = 0;                     // empties / cars of the chunks in front (the same number in every lane)
    float r_min = 0.0f, r_chg = 0.0f, r_max = 0.0f;   // lane 0: the running sums
    long long i_min = 0, i_chg = 0, i_max = 0;
    for (int c = 0; c < chunks; c++) {
        const int slot = (c << 8) + tid;
        const bool valid = slot < S;
        const uint32_t idx = idx0 + (uint32_t) slot;
        float power = 0.0f, t_target = 0.0f, t_soc = 0.0f;
        int tl = 0;
        bool charge = false;
        if (valid) {  // what pass 1 left (this lane's own stores)
            if (MODE == MODE_COMPAT) {
                const u32x4 hot = ((CHUB_G(u32x4)) sl.hot)[idx];
                power = __uint_as_float(hot.x); t_soc = __uint_as_float(hot.z);
                tl = (int) (hot.w & 127u);
                charge = (hot.w & 128u) != 0u;
                if (tl > 0) t_target = tb.ttab[k][hot_level(hot.w)];
            } else {
                const uint32_t w0 = sl.hot[idx];
                tl = ps_tl(w0);
                charge = (w0 & kPsChg) != 0u;
                if (tl > 0) {
                    const f32x2 row = *(CHUB_G(const f32x2)) ((CHUB_G(const char)) cls + ((size_t) ps_cls(w0) * (kClsRow * 8u) + ps_n(w0) * 8u));
                    power = row.x; t_soc = row.y;
                    t_target = tb.ttab[k][ps_lev(w0)];
                }
            }
        }
        bool car = tl > 0;
        const bool empty = valid && !car;
        const uint64_t be = __ballot(empty);
        if (lane == 0) s_ball[wave] = be;
        __syncthreads();
        int chunk_empties = 0, rank = base + prefix_count(be);
        for (int w = 0; w < 4; w++) {
            const int n = __popcll(s_ball[w]);
            chunk_empties += n;
            rank += w < wave ? n : 0;
        }
        if (MODE == MODE_COMPAT) {
            if (tid == 0) {  // the chunk's new cars, ascending slot order == ascending rank
                int n_new = assign - base;
                n_new = n_new < 0 ? 0 : (n_new > chunk_empties ? chunk_empties : n_new);
                for (int rr = 0; rr < n_new; rr++) {
                    s_soc[rr] = arrive_soc_from(rs.normal_d(7.0, 3.0));
                    s_v[rr] = (uint32_t) rs.level();
                    const int late = (int) roundf(rs.normal_f(2.0f, 2.0f));  // mk_late_time("slow"), CHS.hpp:816-830
                    s_w[rr] = (uint32_t) (late < 0 ? 0 : late);
                }
            }
            __syncthreads();
        }
        if (empty && rank < assign) {
            if (MODE == MODE_COMPAT) {
                const int r = rank - base;
                const int lev = (int) s_v[r];
                const float target = uniform_level(lev, 80.0f, 100.0f);
                const NewCar nc = fast ? make_car<0>(s_soc[r], lev, soc_to_time<0>(target, cp), (int) s_w[r], cp)
                                       : make_car<1>(s_soc[r], lev, soc_to_time<1>(target, cp), (int) s_w[r], cp);
                t_target = nc.t_target; t_soc = nc.t_soc; tl = nc.stay; power = nc.power;
                car = tl > 0;
                u32x4 h2;
                h2.x = __float_as_uint(power); h2.y = __float_as_uint(nc.soc); h2.z = __float_as_uint(t_soc);
                h2.w = (uint32_t) tl | ((uint32_t) (nc.stay | (nc.lev << 7)) << 8);
                ((CHUB_G(u32x4)) sl.hot)[idx] = h2;
            } else {
                PhiloxCtx px{hp.key[0], hp.key[1], CHUB_TICK(hp, sa.tick), (uint32_t) (hp.env_id0 + env)};
                const U4 o = px.block(SITE_SOC, (uint32_t) (hub0 + slot), 0);
                const uint32_t cl = o.v[0] >> kSocLevelShift, lev = o.v[1] % 1000u;
                const f32x2 e0 = *(CHUB_G(const f32x2)) ((CHUB_G(const char)) cls + (size_t) cl * (kClsRow * 8u));
                t_target = tb.ttab[k][lev];
                const int late = late_from_word(tb.late_thr, o.v[2]);
                int stay = (int) ceilf(__fsub_rn(t_target, e0.y)) + late;
                stay = stay > kMaxStay ? kMaxStay : stay;
                power = e0.x; t_soc = e0.y; tl = stay;
                car = tl > 0;
                sl.hot[idx] = car ? ps_make(stay, cl, lev) : 0u;
                sl.stay8[idx] = (uint8_t) stay;
            }
        }
        const bool urgent = car && must_charge(t_target, t_soc, tl);
        const uint64_t bc = __ballot(car);
        if (lane == 0) s_ball[4 + wave] = bc;
        if (MODE == MODE_COMPAT) {
            s_a[tid] = urgent ? power : 0.0f;
            s_b[tid] = charge ? power : 0.0f;
            s_c[tid] = car ? power : 0.0f;
        } else {
            const int q = kw_to_fixed(power);
            s_u[tid] = (uint32_t) (urgent ? q : 0);
            s_v[tid] = (uint32_t) (charge ? q : 0);
            s_w[tid] = (uint32_t) (car ? q : 0);
        }
        __syncthreads();
        cars += __popcll(s_ball[4]) + __popcll(s_ball[5]) + __popcll(s_ball[6]) + __popcll(s_ball[7]);
        if (tid == 0) {
            const int n = S - (c << 8) < 256 ? S - (c << 8) : 256;
            if (MODE == MODE_COMPAT) {  // the reference adds the slot powers sequentially in f32 (CHS.hpp:1244-1255)
                for (int i = 0; i < n; i++) {
                    r_max = __fadd_rn(r_max, s_c[i]);
                    r_min = __fadd_rn(r_min, s_a[i]);
                    r_chg = __fadd_rn(r_chg, s_b[i]);
                }
            } else {  // order-independent: 64-bit sums of the slot powers in units of 2^-19 kW, one rounding to f32
                for (int i = 0; i < n; i++) {
                    i_min += (long long) (int) s_u[i];
                    i_chg += (long long) (int) s_v[i];
                    i_max += (long long) (int) s_w[i];
                }
            }
        }
        base += chunk_empties;
        __syncthreads();  // the chunk's LDS areas are free again
    }
    if (tid == 0) {
        if (MODE == MODE_COMPAT) {
            rs.store(ctx->cr, sa.rng_cur, env);
        } else {
            r_min = (float) i_min * (1.0f / 524288.0f);
            r_chg = (float) i_chg * (1.0f / 524288.0f);
            r_max = (float) i_max * (1.0f / 524288.0f);
        }
        rec_store(st.rec, sidx, r_min, r_chg, r_max, pkd_make(s_hdr[2], s_hdr[1], cars));
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// k_slot_packed: the PHILOX step (production).  PHILOX slot state is laid out hub-major, [env][hub slot] (station 0's piles,
// then station 1's), like the action rows: the workgroup's BLOCK * T virtual lanes map onto epb = BLOCK * T / (S0 + S1) whole
// envs laid end to end, so its state loads, action loads and state stores are each ONE contiguous run of memory, read or
// written once.  A unit = one (env, station) = S_k consecutive virtual lanes; units may straddle a wave boundary, so every
// (virtual) wave publishes its ballot of empty slots in LDS (admission rank = empties of the unit in the previous wave +
// empties below the lane in its own wave), and the station sums are integer LDS atomics (order-independent by definition).
// A slot is empty after this step iff its stay is over, which the state word says by itself: the ballots, the barrier,
// the renege / balk / admission arithmetic and the queueing of the admitted lanes all run while the class-row reads are
// still in flight.  The new cars of the workgroup (a handful) and the station records are the last wave's business, one
// lane per car / per unit; the other waves are done after the second barrier.
//
// PackedArgs: everything a wave needs up front, by value in the kernel arguments and requested in one scalar batch.
struct PackedArgs {
    uint32_t S[2], type[2];
    uint32_t n_envs, epb, magic, cls_delta;  // envs per workgroup; 2^20 / (S0 + S1) + 1; byte distance cls[1] - cls[0]
    CHUB_G(uint32_t) state;          // [N][S0 + S1]: one word per slot
    CHUB_G(uint32_t) rec;
    CHUB_G(const uint32_t) pk;       // this step's station draws per unit, decoded (dk_make); RESET: the raw initial-occupancy draws
    CHUB_G(const float) actions;
    CHUB_G(const float) cls0;        // station 0's class table; station 1's is cls_delta bytes further
    CHUB_G(const float) ttab2;       // [2][1024] soc_to_time(target level) of both stations' curves (1000 used)
    CHUB_G(const uint32_t) car_tape; // TAPE: [N][S0 + S1][2] per slot, used if the slot admits a car this step: class, level | late << 16
    // add_car's Philox block (the new cars come last in the workgroup, behind two barriers: nothing of theirs should wait
    // for a load that could have been issued at the start of the wave)
    uint32_t key[2], gid0, tick;     // Philox key, global id of env 0, host tick
    const uint32_t *tick_base;       // device-side tick offset (graph replays), added to tick
    uint32_t late[8];                // the first 8 thresholds of mk_late_time's table (late_from_word); beyond them with probability 0.3 %
    CHUB_G(const uint8_t) env_mask;  // per-env clocks: non-zero = the launch serves this env (null: every env)
    uint32_t blk0;                   // ... and the first workgroup of the range of envs it names (the grid covers that range only)
    uint32_t xcd;                    // non-zero: tiles in XCD-aware order (xcd_order; handles whose streams are cache-resident: the small tile) -- the
                                     // launch's number of workgroups, here so that it arrives with the first batch of arguments (gridDim.x is a load of its own)
    CHUB_G(float) tail_act;          // [N][2] out: the env's two tail actions, for the tail kernel (StationArrays::tail_act)
    CHUB_G(uint8_t) stay8;           // [N][S0 + S1] out, per admitted car: its stay_time (introspection only)
};

typedef const uint32_t __attribute__((address_space(4))) *chub_sptr;  // constant address space: scalar loads
__device__ __forceinline__ uint32_t sload_u32(const void *p, int i) { return ((chub_sptr) (uintptr_t) p)[i]; }

// RESET: evs_reset (CHS.hpp:1209-1231 / 1520-1542) on the same layout: no state comes in, the unit's initial occupancy was drawn
// by k_reset_levels, every wave helps with the (many) new cars.  BIG: a station with more than 64 piles -- a unit then spans
// several waves (its empties are counted over all of them) and its power sums need 64 bits.
// The action rows are read exactly once: CHUB_ACT_NT = 1 loads them with the non-temporal hint, so that they do not push the slot state --
// which the next step reads again -- out of the Infinity Cache (C5: 69 MB of rows per step next to 134 MB of state)
#ifndef CHUB_ACT_NT
#define CHUB_ACT_NT 0
#endif
#if CHUB_ACT_NT
#define CHUB_ACT_LOAD(p) __builtin_nontemporal_load(p)
#else
#define CHUB_ACT_LOAD(p) (*(p))
#endif
#ifndef CHUB_ACC_COPIES
#define CHUB_ACC_COPIES 2  // copies of a unit's LDS accumulators (lanes spread over them by lane number: fewer same-address atomics; 1 / 2 / 4: 21.28 / 20.95 / 21.12 us)
#endif
constexpr int kAccCopies = CHUB_ACC_COPIES;
#ifndef CHUB_EPI_ALL
#define CHUB_EPI_ALL 1  // 1: every wave of the workgroup takes its share of the new cars (a third barrier: -0.2 us at C4); 0: the last wave alone
#endif
// The station records of a workgroup of the packed kernel, one lane of ONE wave per unit (the caller has passed the workgroup's last
// barrier: every car's LDS atomics are in).  FUSED: also into s_rec, for the tails that follow in the same workgroup.
#define CHUB_AT(T_, base, byte_off) (*(CHUB_G(T_)) ((CHUB_G(char)) (base) + (uint32_t) (byte_off)))
template <int BLOCK, int T, bool RESET, bool BIG, bool MASKED, bool FUSED, bool BITS>
__device__ __forceinline__ void packed_records(const PackedArgs &pa, const uint32_t block_local, int *s_acc, const uint32_t *s_unit, u32x4 *s_rec) {
    const int lane = threadIdx.x & 63;
    const int S0 = (int) pa.S[0], S1 = (int) pa.S[1], St = S0 + S1;
    const int epb = (int) pa.epb, N = (int) pa.n_envs;
    const int env_first = (int) block_local * epb;
    const long long *s_acc64 = (const long long *) s_acc;
    // the wave's own LDS atomics above are in program order with the reads below; nothing else touches s_acc any more
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (int i = lane; i < 2 * epb; i += 64) {  // one station record per unit
        const int e = i >> 1, k = i & 1;
        const int env = env_first + e;
        if (env >= N) continue;
        if (MASKED && pa.env_mask[env] == 0) continue;
        const uint32_t su = (uint32_t) (k ? N : 0) + (uint32_t) env;
        if (!RESET && !FUSED && !BITS && k == 0) {  // the env's tail actions sit in the lines this workgroup has just read: hand them to the tail kernel packed
            typedef float f32x2_ __attribute__((ext_vector_type(2)));
            const uint32_t ai = ((uint32_t) env * (uint32_t) (St + 2) + (uint32_t) St) << 2;
            const f32x2_ tv = {CHUB_AT(const float, pa.actions, ai), CHUB_AT(const float, pa.actions, ai + 4u)};
            CHUB_AT(f32x2_, pa.tail_act, (uint32_t) env << 3) = tv;
        }
        uint32_t lf = s_unit[i];
        if ((k ? S1 : S0) == 0) {
            // a station without piles still queues, reneges and balks (receive_car runs on it as on any other, CHS.hpp:1272-1316 /
            // 1583-1627; nobody is ever admitted): its queue length and arrival count move on here, power sums and cars stay 0
            const uint32_t pk = CHUB_AT(const uint32_t, pa.pk, su << 2);
            const bool fast = (k ? pa.type[1] : pa.type[0]) == 0;
            int want, fl;
            if (RESET) {  // evs_reset of a station without piles: init_station_car_number(0, 3) arrivals, nobody queues
                fl = fast ? (int) (int16_t) (pk & 0xFFFFu) : (int) ((pk >> 16) & 0xFFFFu);
                want = fl;
            } else {
                want = dk_want(pk);
                fl = dk_flow(pk);
            }
            // assign_car (CHS.hpp:417-430) with no empty slot: min(line + flow, 0) cars are assigned, the rest queue
            const int as = want < 0 ? want : 0;
            int ln = want - as;
            ln = ln < kMaxLine ? ln : kMaxLine;
            lf = pkd_make(ln, fl, 0);
        }
        u32x4 rv;
        if (!BIG) {
            int a0 = 0, a1 = 0, a2 = 0, a3 = 0;
#pragma unroll
            for (int c = 0; c < kAccCopies; c++) {
                const int *acc = s_acc + 4 * i + c * 8 * epb;
                a0 += acc[0]; a1 += acc[1]; a2 += acc[2]; a3 += acc[3];
            }
            rv = u32x4{__float_as_uint(fixed_to_kw(a0)), __float_as_uint(fixed_to_kw(a1)), __float_as_uint(fixed_to_kw(a2)),
                       lf | ((uint32_t) a3 << 16)};
        } else {
            const long long *a64 = s_acc64 + 4 * i;
            rv = u32x4{__float_as_uint((float) a64[0] * (1.0f / 524288.0f)), __float_as_uint((float) a64[1] * (1.0f / 524288.0f)),
                       __float_as_uint((float) a64[2] * (1.0f / 524288.0f)), lf | ((uint32_t) a64[3] << 16)};
        }
        CHUB_AT(u32x4, pa.rec, su << 4) = rv;
        if (FUSED) s_rec[i] = rv;
    }
}
#undef CHUB_AT

// MASKED: per-env clocks (the launch serves the envs of a mask); the lock-step instantiation carries none of it.
// FUSED (k_step_fused, small batches): the workgroup goes on by itself -- the body then returns IN FRONT of its third barrier with what
// this wave does next (w + 1: wave w < last; WAVES: the last wave, which will write the records, also into s_rec, between the two
// halves of its tails), does not hand the tail actions over through memory, and calls hook.prefetch() behind its first loads /
// hook.park() in front of its first barrier (the tail's table rows travel with the slot loads).
struct NoHook {
    __device__ __forceinline__ void prefetch() {}
    __device__ __forceinline__ void park() {}
};
// BITS: the pile decisions arrive as one bit per pile ([N][ceil(S / 64)] u64 behind pa.actions) instead of a row of floats:
// 8 bytes per env and word instead of 4 per pile (chub_step_bits; the tail kernel reads the two tail actions from their own array).
// PIPED (k_steps_piped): the new cars are the business of the upper half of the workgroup's slot waves -- the lower half returns right behind the
// third barrier and makes the next step's draws meanwhile.
template <int BLOCK, int T, bool TAPE, bool RESET, bool BIG, bool MASKED, bool FUSED, typename Hook, bool BITS = false, bool PIPED = false>
__device__ __forceinline__ int slot_body_packed(const HubParams &hp, const StepArgs &sa, PackedArgs &pa, const Tables &tb,
                                                const uint32_t block_local, uint32_t *q_cnt, uint32_t *q_new,
                                                uint64_t *s_ball, int *s_acc, uint32_t *s_unit, Hook &hook, u32x4 *s_rec, uint32_t *s_uinfo) {
    // T slots per lane: virtual lane v = tid + j * BLOCK (j < T), virtual wave = wave + j * (BLOCK / 64).  All T slots' loads
    // are in flight together and the barriers are shared.
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    constexpr int WAVES = BLOCK / 64;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int S0 = (int) pa.S[0], S1 = (int) pa.S[1], St = S0 + S1;
    const int epb = (int) pa.epb, N = (int) pa.n_envs;
    const int env_first = (int) block_local * epb;
    const uint32_t idx0 = (uint32_t) env_first * (uint32_t) St;
    CHUB_STAMP_DECL(16);
    CHUB_STAMP_REAL(12);
    CHUB_STAMP(0);
    // every array reached from here is < 4 GiB (checked at create), so addresses are a uniform base + a 32-bit byte offset
    // per lane: the loads and stores take the base from SGPRs and need no 64-bit address arithmetic
#define CHUB_AT(T_, base, byte_off) (*(CHUB_G(T_)) ((CHUB_G(char)) (base) + (uint32_t) (byte_off)))

    // phase A loads: slot state, action, the unit's queue length and packed draws.  No zero fill for the lanes without a slot:
    // what they hold is never used (their occupancy is forced to 0 below).
    int e_[T], k_[T], slot[T];
    bool valid[T];
    uint32_t sidx[T];
    uint32_t s2[T];
    float act[T];
    u32x2 actw[T];  // BITS: the word of the env's decision bits this slot's bit sits in
    int hs_[T];
    uint32_t pk_in[T];
#pragma unroll
    for (int j = 0; j < T; j++) {
        const int v = tid + j * BLOCK;
        // v / (S0 + S1) (magic = 2^20 / St + 1, checked on the host); 24-bit multiplies (full rate; v < 2^11, magic <= 2^20 + 1, the
        // product below 2^32): the 32-bit v_mul_lo_u32 is a quarter-rate instruction
        e_[j] = (int) (__umul24((uint32_t) v, pa.magic) >> 20);
        const int hs = v - (int) __umul24((uint32_t) e_[j], (uint32_t) St);  // hub slot
        k_[j] = hs >= S0 ? 1 : 0;
        hs_[j] = hs;
        slot[j] = hs - (k_[j] ? S0 : 0);
        const int env = env_first + e_[j];
        valid[j] = e_[j] < epb && env < N;
        uint32_t served = 1u;  // per-env clocks: is the env served by this launch?  Requested with the state, looked at after it
        if (MASKED && valid[j]) served = pa.env_mask[env];
        sidx[j] = (uint32_t) (k_[j] ? N : 0) + (uint32_t) env;
        asm volatile("" : "=v"(s2[j]), "=v"(act[j]), "=v"(pk_in[j]), "=v"(actw[j]));
        if (RESET) {
            s2[j] = 0u;
            act[j] = 0.0f;
        }
        if (valid[j]) {
            if (!RESET) {
                s2[j] = CHUB_AT(uint32_t, pa.state, (idx0 + (uint32_t) v) << 2);
                if (BITS) actw[j] = CHUB_AT(const u32x2, pa.actions, ((uint32_t) env * (((uint32_t) St + 63u) >> 6) + ((uint32_t) hs >> 6)) << 3);
                else act[j] = CHUB_ACT_LOAD((CHUB_G(const float)) ((CHUB_G(const char)) pa.actions + (uint32_t) ((idx0 + (uint32_t) v + 2u * (uint32_t) env) << 2)));  // row stride S0 + S1 + 2
            }
            pk_in[j] = CHUB_AT(const uint32_t, pa.pk, sidx[j] << 2);
        }
        if (MASKED && valid[j]) valid[j] = served != 0u;
    }
    if (!TAPE) {  // the new cars' Philox inputs: scalar registers from here on, requested behind the first loads
        pa.tick += sload_u32(pa.tick_base, 0);  // CHUB_TICK
        asm volatile("" : "+s"(pa.key[0]), "+s"(pa.key[1]), "+s"(pa.gid0), "+s"(pa.tick), "+s"(pa.late[0]), "+s"(pa.late[1]), "+s"(pa.late[2]),
                          "+s"(pa.late[3]), "+s"(pa.late[4]), "+s"(pa.late[5]), "+s"(pa.late[6]), "+s"(pa.late[7]));
    }
    CHUB_STAMP(1);  // first loads issued
    if (FUSED) hook.prefetch();
    if (tid == 0) q_cnt[0] = 0;
    for (int i = tid; i < (BIG ? 16 : 8 * kAccCopies) * epb; i += BLOCK) s_acc[i] = 0;
    long long *s_acc64 = (long long *) s_acc;  // BIG: {min, charge, max power, cars} as four 64-bit sums per unit

    // ---- second round trip, only for the slots whose car stays: where it is on its curve (entries n, n + 1 of its class
    // row) and the time its target SoC stands for (4 bytes of a table that sits in the vector L1), both in flight together.  A
    // car that leaves this step is wiped right after its car_step (CHS.hpp:1196-1201): nothing of it is needed, and a slot is
    // empty after remove_car (CHS.hpp:912-923 / 1077-1088) iff it had at most one slot of stay left
    uint32_t w0[T];
    int tl[T];
    bool stays[T];
    f32x4 row[T];
    float ttg[T];
#pragma unroll
    for (int j = 0; j < T; j++) {
        asm volatile("" : "+v"(s2[j]));
        w0[j] = valid[j] ? s2[j] : 0u;
        tl[j] = ps_tl(w0[j]);
        stays[j] = tl[j] > 1;
        asm volatile("" : "=v"(row[j]), "=v"(ttg[j]));
        if (stays[j]) {
            row[j] = CHUB_AT(const f32x4, pa.cls0, (k_[j] ? pa.cls_delta : 0u) + (ps_cls(w0[j]) << 8) + (ps_n(w0[j]) << 3));
            ttg[j] = CHUB_AT(const float, pa.ttab2, (k_[j] ? 4096u : 0u) + (ps_lev(w0[j]) << 2));
        }
    }

    CHUB_STAMP(2);  // state words here, class-row / target-time reads issued
    // ---- receive_car (CHS.hpp:1272-1316 / 1583-1627) while those are in flight
    bool empty[T];
    uint64_t be[T];
#pragma unroll
    for (int j = 0; j < T; j++) {
        empty[j] = valid[j] && !stays[j];
        be[j] = __ballot(empty[j]);
        if (lane == 0) s_ball[wave + j * WAVES] = be[j];
    }
    if (FUSED) hook.park();
    CHUB_STAMP(3);
    __syncthreads();
    CHUB_STAMP(4);  // barrier 1 passed
    // ---- every unit's empties, ONCE per unit: a lane per unit reads the ballots of the (at most two) virtual waves the unit lies in
    // and leaves, in one word, the empties in front of the unit inside its first wave, the unit's empties in that wave, its empties in
    // all, and which wave that is.  (Rounds 2-3 had every LANE work its unit's 64-bit masks out for itself, two popcounts and an LDS
    // read each: a third of the kernel's vector instructions.)  Stations of more than 64 piles keep that form below.
    if (!BIG) {
        for (int u = tid; u < 2 * epb; u += BLOCK) {
            const int e = u >> 1, k = u & 1;
            const int Sk = k ? S1 : S0;
            const int U0 = e * St + (k ? S0 : 0), U1 = U0 + Sk;  // the unit's virtual lanes [U0, U1) of the workgroup
            const int wv = U0 >> 6, lo = U0 & 63;
            const int hi0 = (U1 - (wv << 6)) < 64 ? (U1 - (wv << 6)) : 64;
            uint32_t info = (uint32_t) wv << 24;
            if (Sk > 0) {
                const uint64_t b0 = s_ball[wv];
                const uint64_t below_lo = lo ? (~0ull >> (64 - lo)) : 0ull, below_hi = ~0ull >> (64 - hi0);
                const int start_e = __popcll(b0 & below_lo), cnt0 = __popcll(b0 & below_hi & ~below_lo);
                const int over = U1 - ((wv + 1) << 6);  // lanes of the unit in the next virtual wave
                const int cnt1 = over > 0 ? __popcll(s_ball[wv + 1] & (~0ull >> (64 - over))) : 0;
                info |= (uint32_t) start_e | ((uint32_t) cnt0 << 8) | ((uint32_t) (cnt0 + cnt1) << 16);
            }
            s_uinfo[u] = info;
        }
    }

    // ---- the occupied slots' step: urgency, feasibility / on-off, car_step, departure (CHS.hpp:1188-1202 / 1499-1513) and
    // their share of calculate_output (CHS.hpp:1233-1261 / 1544-1572) -- nothing here waits for the admission
    uint32_t w0n[T];
#pragma unroll
    for (int j = 0; j < T; j++) {
        if (BITS) asm volatile("" : "+v"(row[j]), "+v"(ttg[j]), "+v"(actw[j]));
        else asm volatile("" : "+v"(row[j]), "+v"(ttg[j]), "+v"(act[j]));
        const int u = 2 * e_[j] + k_[j];
        int *acc = s_acc + 4 * u + (BIG ? 0 : (lane & (kAccCopies - 1)) * 8 * epb);  // the unit's {min, charge power | max power, cars}: two 64-bit LDS atomics per car
        w0n[j] = 0u;
        if (stays[j]) {
            const float t_target = ttg[j];
            // action_to_real (MGR:384-393), judge_feasibility (CHS.hpp:1404-1413)
            const bool bit_on = BITS && ((((hs_[j] & 32) ? actw[j].y : actw[j].x) >> (hs_[j] & 31)) & 1u) != 0u;
            const bool on = (BITS ? bit_on : act[j] >= kActOnThreshold) || must_charge(t_target, row[j].y, tl[j]);
            const float power = on ? row[j].z : row[j].x, t_soc = on ? row[j].w : row[j].y;  // car_step = the next entry of the class row
            const int q = kw_to_fixed(power);
            const bool urgent = must_charge(t_target, t_soc, tl[j] - 1);
            w0n[j] = (w0[j] & ~kPsChg) - 1u + (on ? (kPsStep | kPsChg) : 0u);
            if (!BIG) {
                atomicAdd((unsigned long long *) (acc + 2), (unsigned long long) (uint32_t) q | (1ull << 32));
                if (on || urgent)
                    atomicAdd((unsigned long long *) acc,
                              (unsigned long long) (urgent ? (uint32_t) q : 0u) | ((unsigned long long) (on ? (uint32_t) q : 0u) << 32));
            } else {
                unsigned long long *a64 = (unsigned long long *) (s_acc64 + 4 * u);
                atomicAdd(a64 + 2, (unsigned long long) q);
                atomicAdd(a64 + 3, 1ull);
                if (urgent) atomicAdd(a64 + 0, (unsigned long long) q);
                if (on) atomicAdd(a64 + 1, (unsigned long long) q);
            }
        }
    }
    CHUB_STAMP(5);  // rows consumed, sums added
    if (!BIG) __syncthreads();  // the units' words are in
    CHUB_STAMP(6);  // barrier 2 passed

    // ---- receive_car (CHS.hpp:1272-1316 / 1583-1627): admission
    int line[T], flow[T];
    bool adm[T];
    uint64_t ba[T];
    uint32_t n_push = 0;
#pragma unroll
    for (int j = 0; j < T; j++) {
        const int vw = wave + j * WAVES;
        int empties, rank;
        if (!BIG) {
            // rank = the unit's empties in front of this lane: inside the unit's first wave, empties of the wave below the lane minus
            // those in front of the unit; in its second wave, the unit's empties of the first wave + those below the lane
            const uint32_t info = valid[j] ? s_uinfo[2 * e_[j] + k_[j]] : 0u;  // (lanes past the workgroup's last env hold no unit)
            const int m = prefix_count(be[j]);
            empties = (int) ((info >> 16) & 255u);
            rank = (int) (info >> 24) == vw ? m - (int) (info & 255u) : m + (int) ((info >> 8) & 255u);
        } else {
            const int Sk = k_[j] ? S1 : S0;
            const int ub = e_[j] * St + (k_[j] ? S0 : 0) - vw * 64, ue = ub + Sk;  // relative to this wave's lane 0
            // a unit of up to 256 lanes: its empties in every virtual wave it touches; those in earlier waves come before this lane
            empties = 0;
            rank = 0;
#pragma unroll
            for (int w = 0; w < WAVES * T; w++) {
                int lo = ub + (vw - w) * 64, hi = ue + (vw - w) * 64;  // the unit's lanes relative to wave w's lane 0
                lo = lo < 0 ? 0 : lo;
                hi = hi > 64 ? 64 : hi;
                if (hi > lo) {
                    const uint64_t m = (~0ull >> (64 - hi)) & (~0ull << lo);
                    const uint64_t b = (w == vw ? be[j] : s_ball[w]) & m;
                    empties += __popcll(b);
                    if (w < vw) rank += __popcll(b);
                    else if (w == vw) rank += prefix_count(b);
                }
            }
        }
        line[j] = 0;
        flow[j] = 0;
        int assign = 0;
        if (valid[j]) {
            const uint32_t pk = pk_in[j];
            int want;
            if (RESET) {
                // the unit's initial occupancy, drawn by k_reset_levels just before.  The fast station records the raw draw, which is
                // negative for small stations (mu - 3 < 0, CHS.hpp:1617, 832-842): assign_car then admits nobody, the queue stays empty
                const bool fast = (k_[j] ? pa.type[1] : pa.type[0]) == 0;
                flow[j] = fast ? (int) (int16_t) (pk & 0xFFFFu) : (int) ((pk >> 16) & 0xFFFFu);
                want = flow[j];
            } else {
                // queue after the renege pass + the arrivals that stay, and flow_in: decoded one launch ahead, once per unit (dk_make)
                want = dk_want(pk);
                flow[j] = dk_flow(pk);
            }
            assign = want < empties ? want : empties;  // assign_car, CHS.hpp:417-430
            line[j] = want - assign;
            line[j] = line[j] < kMaxLine ? line[j] : kMaxLine;
        }
        adm[j] = empty[j] && rank < assign;
        ba[j] = __ballot(adm[j]);
        n_push += (uint32_t) __popcll(ba[j]);
    }
    if (n_push) {  // wave-uniform
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(&q_cnt[0], n_push);
        base = __shfl(base, 0);
#pragma unroll
        for (int j = 0; j < T; j++) {
            if (adm[j]) q_new[base + prefix_count(ba[j])] = (uint32_t) (tid + j * BLOCK);
            base += (uint32_t) __popcll(ba[j]);
        }
    }
#pragma unroll
    for (int j = 0; j < T; j++) {
        if (valid[j] && !adm[j]) CHUB_AT(uint32_t, pa.state, (idx0 + (uint32_t) (tid + j * BLOCK)) << 2) = w0n[j];
        if (valid[j] && slot[j] == 0) s_unit[2 * e_[j] + k_[j]] = pkd_make(line[j], flow[j], 0);
    }
    CHUB_STAMP(7);  // admission done, state stores issued
    __syncthreads();
    CHUB_STAMP(8);  // barrier 3 passed
    // Everything left -- the workgroup's new cars and, after them, the station records -- is the last wave's business: the
    // other waves are done (their wave slots go to the next workgroup instead of idling at a third barrier through the
    // Philox block and the dependent table read of the new cars).  RESET: about half of all slots get a car, so every wave
    // serves its share and the workgroup meets again in front of the records.
    constexpr bool ALL = RESET || CHUB_EPI_ALL || FUSED;  // every wave takes new cars
    if (!ALL && wave != WAVES - 1) return 0;
    if (PIPED && wave < WAVES / 2) return wave + 1;
    // ---- add_car (CHS.hpp:864-877 / 1029-1042) for the workgroup's new cars, one lane per car
    const uint32_t n_adm = q_cnt[0];
    for (uint32_t i = (uint32_t) (PIPED ? tid - (WAVES / 2) * 64 : (ALL ? tid : lane)); i < n_adm; i += (PIPED ? BLOCK - (WAVES / 2) * 64 : (ALL ? BLOCK : 64))) {
        const int src = (int) q_new[i];
        const int s_e = (int) (__umul24((uint32_t) src, pa.magic) >> 20);
        const int s_hs = src - (int) __umul24((uint32_t) s_e, (uint32_t) St);
        const int s_k = s_hs >= S0 ? 1 : 0;
        uint32_t c, lev;
        int late;
        if (TAPE) {
            const u32x2 tp = CHUB_AT(const u32x2, pa.car_tape, (idx0 + (uint32_t) src) << 3);
            c = tp.x;
            lev = tp.y & 0xFFFFu;
            late = (int) (tp.y >> 16);
        } else {
            PhiloxCtx p2{pa.key[0], pa.key[1], pa.tick, pa.gid0 + (uint32_t) (env_first + s_e)};
            const U4 o = p2.block(SITE_SOC, (uint32_t) s_hs, 0);  // word 0 SoC class, 1 target level, 2 extra stay
            c = o.v[0] >> kSocLevelShift;
            lev = o.v[1] % 1000u;
            late = 0;  // late_from_word, the first 8 thresholds from registers
#pragma unroll
            for (int j = 0; j < 8; j++) late += (o.v[2] >= pa.late[j]) ? 1 : 0;
            if (__any(o.v[2] >= pa.late[7])) {
#pragma unroll
                for (int j = 8; j < 16; j++) late += (o.v[2] >= tb.late_thr[j]) ? 1 : 0;
            }
        }
        f32x2 e0 = CHUB_AT(const f32x2, pa.cls0, (s_k ? pa.cls_delta : 0u) + (c << 8));
        float tt_ = CHUB_AT(const float, pa.ttab2, (s_k ? 4096u : 0u) + (lev << 2));  // soc_to_time(target), CHS.hpp:867 / 1032
        asm volatile("" : "+v"(e0), "+v"(tt_));  // both lookups in flight together
        int st_ = (int) ceilf(__fsub_rn(tt_, e0.y)) + late;  // calculate_min_charging_time + mk_late_time
        st_ = st_ > kMaxStay ? kMaxStay : st_;
        CHUB_AT(uint32_t, pa.state, (idx0 + (uint32_t) src) << 2) = st_ > 0 ? ps_make(st_, c, lev) : 0u;
        CHUB_AT(uint8_t, pa.stay8, idx0 + (uint32_t) src) = (uint8_t) st_;
        if (st_ > 0) {
            const int q = kw_to_fixed(e0.x);
            if (!BIG) {
                int *ac = s_acc + 4 * (2 * s_e + s_k) + (lane & (kAccCopies - 1)) * 8 * epb;
                atomicAdd((unsigned long long *) (ac + 2), (unsigned long long) (uint32_t) q | (1ull << 32));
                if (must_charge(tt_, e0.y, st_)) atomicAdd(ac, q);
            } else {
                unsigned long long *a64 = (unsigned long long *) (s_acc64 + 4 * (2 * s_e + s_k));
                atomicAdd(a64 + 2, (unsigned long long) q);
                atomicAdd(a64 + 3, 1ull);
                if (must_charge(tt_, e0.y, st_)) atomicAdd(a64 + 0, (unsigned long long) q);
            }
        }
    }
    CHUB_STAMP(9);  // new cars done
    // FUSED (k_step_fused): the workgroup's third barrier and the station records are the caller's business from here (its last wave
    // runs the first half of its envs' tails in front of them): every wave returns its role, the last wave WAVES
    if (FUSED) return wave == WAVES - 1 ? WAVES : wave + 1;
    if (ALL) {
        __syncthreads();
        CHUB_STAMP(10);  // barrier 4 passed
        CHUB_STAMP_REAL(13);
#if CHUB_TRACE
        if (sa.stamps_slot && tid == 0) {
            for (int i = 0; i <= 10; i++) sa.stamps_slot[(size_t) block_local * 16 + i] = stamp_[i];
            sa.stamps_slot[(size_t) block_local * 16 + 12] = stamp_[12];
            sa.stamps_slot[(size_t) block_local * 16 + 13] = stamp_[13];
        }
#endif
        if (wave != WAVES - 1) return 0;
    }
    packed_records<BLOCK, T, RESET, BIG, MASKED, false, BITS>(pa, block_local, s_acc, s_unit, nullptr);
#if CHUB_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    CHUB_STAMP(11);  // records stored: the workgroup is done
    if (sa.stamps_slot && lane == 0) sa.stamps_slot[(size_t) block_local * 16 + 11] = stamp_[11];
#endif
    return 0;
#undef CHUB_AT
}

template <int BLOCK, int T, bool TAPE, bool RESET, bool BIG, bool MASKED, bool BITS = false>
__global__ __launch_bounds__(BLOCK, (BLOCK <= 256 ? 8 : 2048 / BLOCK)) void k_slot_packed(const DevCtx *__restrict__ ctx, StepArgs sa, PackedArgs pa_in) {
    __shared__ uint32_t q_new[BLOCK * T];
    __shared__ uint32_t q_cnt[2];
    __shared__ uint64_t s_ball[BLOCK * T / 64 + 2];                   // [1 + virtual wave]: a unit's neighbouring wave may be "wave -1" or "wave WAVES"
    __shared__ __attribute__((aligned(16))) int s_acc[2 * BLOCK * T * kAccCopies];  // 2 epb <= BLOCK * T / 2 units x {min, charge, max power, cars} (BIG: as 64-bit sums, few units)
    __shared__ uint32_t s_unit[BLOCK * T / 2];                        // per unit: line | flow << 8
    __shared__ uint32_t s_uinfo[BLOCK * T / 2];                       // per unit: where its empties are (the admission's unit pass)
    // all kernel arguments this wave needs, requested in ONE batch of scalar loads at its very start (the single asm
    // statement makes every one of them live here), instead of in a chain of dependent loads in front of the first
    // vector load
    PackedArgs pa = pa_in;
    asm volatile("" : "+s"(pa.S[0]), "+s"(pa.S[1]), "+s"(pa.type[0]), "+s"(pa.type[1]), "+s"(pa.n_envs), "+s"(pa.epb), "+s"(pa.magic),
                      "+s"(pa.cls_delta), "+s"(pa.state), "+s"(pa.rec), "+s"(pa.pk), "+s"(pa.actions), "+s"(pa.cls0), "+s"(pa.ttab2));
    NoHook hook;
    (void) slot_body_packed<BLOCK, T, TAPE, RESET, BIG, MASKED, false, NoHook, BITS>(ctx->hp, sa, pa, ctx->tb, xcd_order(blockIdx.x, pa.xcd, MASKED ? 0u : pa.xcd) + (MASKED ? pa.blk0 : 0u), q_cnt, q_new,
                                                                                     s_ball + 1, s_acc, s_unit, hook, nullptr, s_uinfo);
}

// What the tail's first loads need, by value in the kernel arguments: one scalar load at the start of the wave instead
// of a dozen dependent hops through the context pointer in front of the load burst of this latency-bound kernel.
struct TailArgs {
    CHUB_G(const double) ou;
    CHUB_G(const double) price_noise;
    CHUB_G(const double) cap;
    CHUB_G(const int16_t) pv_day;
    CHUB_G(const int16_t) wd_day;
    CHUB_G(const uint8_t) q_len;
    CHUB_G(const uint8_t) hv_line;
    CHUB_G(const uint32_t) drw;   // this step's pre-drawn env variates
    CHUB_G(const uint8_t) drw_cnt;
    CHUB_G(uint32_t) rec;
    CHUB_G(const float) actions;
    uint32_t n_envs, act_dim, s_tot, xcd;  // xcd: the tail workgroups in XCD-aware order (xcd_order), as the slot kernel's tiles
    // the rows of this slot of the day and what the tail's Philox context needs: nothing in front of the load burst goes
    // through the context pointer (each hop there is a dependent scalar round trip of this latency-bound kernel)
    CHUB_G(const float) tail_act;  // [N][2] the tail actions as the packed slot kernel left them, or null: read the action rows
    CHUB_G(const double) pv_row;   // pvT + t_next * 100
    CHUB_G(const double) wd_row;   // wdT + t_next * 150
    CHUB_G(const double) pv_row_now;  // ... and the rows of the slot being simulated: what the previous make_state looked up
    CHUB_G(const double) wd_row_now;
    CHUB_G(const double) hy_table;
    const uint32_t *tick_base;
    double sin_t;                  // sin96[t_next] (lock-step; per-env clocks read the table)
    uint32_t key[2], gid0, pad2;
};
inline TailArgs make_tail_args(const EnvArrays &ev, const StationArrays &st, const HubParams &hp, const StepArgs &sa, const PackedPtrs &pp,
                               bool reset) {
    TailArgs ta;
    const int t_next = reset ? 0 : (sa.t + 1) % 96;
    ta.tail_act = (hp.rng_mode == MODE_PHILOX && hp.packed && !sa.load_mode && !reset) ? (CHUB_G(const float)) st.tail_act : nullptr;
    if (sa.act_bits && !reset) ta.tail_act = (CHUB_G(const float)) sa.act_tail;  // packed actions: the caller's [N][2] array is that already
    ta.pv_row = (CHUB_G(const double)) (pp.tb->pvT + t_next * 100);
    ta.wd_row = (CHUB_G(const double)) (pp.tb->wdT + t_next * 150);
    ta.pv_row_now = (CHUB_G(const double)) (pp.tb->pvT + (reset ? 0 : sa.t) * 100);
    ta.wd_row_now = (CHUB_G(const double)) (pp.tb->wdT + (reset ? 0 : sa.t) * 150);
    ta.hy_table = (CHUB_G(const double)) pp.tb->hy_table;
    ta.tick_base = hp.tick_base;
    ta.sin_t = pp.sin96[t_next];
    ta.key[0] = hp.key[0];
    ta.key[1] = hp.key[1];
    ta.gid0 = (uint32_t) hp.env_id0;
    ta.pad2 = 0;
    ta.ou = (CHUB_G(const double)) ev.ou;
    ta.price_noise = (CHUB_G(const double)) ev.price_noise;
    ta.cap = (CHUB_G(const double)) ev.cap;
    ta.pv_day = (CHUB_G(const int16_t)) ev.pv_day;
    ta.wd_day = (CHUB_G(const int16_t)) ev.wd_day;
    ta.q_len = (CHUB_G(const uint8_t)) ev.q_len;
    ta.hv_line = (CHUB_G(const uint8_t)) ev.hv_line;
    ta.drw = (CHUB_G(const uint32_t)) ev.drw[sa.tick & 1u];
    ta.drw_cnt = (CHUB_G(const uint8_t)) ev.drw_cnt[sa.tick & 1u];
    ta.rec = (CHUB_G(uint32_t)) st.rec;
    ta.actions = (CHUB_G(const float)) sa.actions;
    ta.n_envs = (uint32_t) hp.n_envs;
    ta.act_dim = (uint32_t) hp.act_dim;
    ta.s_tot = (uint32_t) (hp.S[0] + hp.S[1]);
    ta.xcd = (hp.rng_mode == MODE_PHILOX && hp.packed && hp.xcd) ? 1u : 0u;  // (COMPAT tails in this order: measured, no gain)
    return ta;
}

template <bool RESET, int MODE, int BLOCK>
__global__ __launch_bounds__(BLOCK, 7) void k_slot(const DevCtx *__restrict__ ctx, StepArgs sa, int64_t nb0) {
    const HubParams &hp = ctx->hp;
    __shared__ float lds_f[BLOCK];
    __shared__ uint32_t lds_u[2 * BLOCK];
    int k;
    int64_t bl;
    const int64_t bid = blockIdx.x;
    if (sa.station_filter >= 0) {
        k = sa.station_filter;
        bl = bid;
    } else {
        k = (bid >= nb0) ? 1 : 0;
        bl = k ? bid - nb0 : bid;
    }
    if (MODE == MODE_PHILOX) slot_body_wave<RESET, BLOCK>(hp, sa, ctx->sl, ctx->st, ctx->tb, k, bl, lds_f, lds_u);
    else if (hp.type[k] == 0) slot_body_compat<0, RESET, BLOCK>(hp, sa, ctx->sl, ctx->st, ctx->cr, ctx->tb, k, bl, lds_f, lds_u);
    else slot_body_compat<1, RESET, BLOCK>(hp, sa, ctx->sl, ctx->st, ctx->cr, ctx->tb, k, bl, lds_f, lds_u);
}

// ---------------------------------------------------------------------------------------- COMPAT, the split step
// One kernel per station with the unit's first lane walking the env's streams keeps 2 of a wave's 64 lanes busy through the longest
// part of the step (every polar normal of the reference's std::normal_distribution costs a lane about a microsecond): 250 of the
// 279 us of a step at 65 536 envs.  Large batches therefore run
//   (k_compat_empties lane = slot: how many slots of each unit are empty once this step's departures are out -- a slot is empty after
//                     remove_car iff it had at most one slot of stay left, CHS.hpp:912-923 / 1077-1088 -- all the walk needs of the slots;
//                     only where the previous pass was not a split one: k_slot_split leaves the counts for the next step itself)
//   k_compat_walk     lane = ENV, 64 walks per wave: station 0's draws, then station 1's, in the reference's consumption order
//                     (receive_car, CHS.hpp:1272-1316 / 1583-1627; the forecourt's follow in the tail kernel as before), leaving per
//                     unit flow / cars admitted / queue and per admitted car its three variates
//   k_slot_split      lane = slot, both stations in ONE launch: everything else (slot_body_compat<.., SPLIT>)
// Same streams, same order, same arithmetic as the one-kernel-per-station form (chub_options.slot_kernel = 1 keeps that one:
// the parity cross-check; handles of a few envs run k_compat_small).
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_compat_empties(const DevCtx *__restrict__ ctx, StepArgs sa, int64_t nb0) {
    const HubParams &hp = ctx->hp;
    const int64_t bid = blockIdx.x;
    const int k = (bid >= nb0) ? 1 : 0;
    const int64_t bl = k ? bid - nb0 : bid;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int U = hp.U[k], S = hp.S[k];  // (the unit layout of slot_body_compat)
    const int upw = 64 / U, uiw = lane / U, slot = lane - uiw * U;
    const int env = (int) bl * ((BLOCK / 64) * upw) + wave * upw + uiw;
    const bool unit_ok = uiw < upw && env < (int) hp.n_envs && in_group(sa, env);
    const bool valid = unit_ok && slot < S;
    const uint64_t unit_mask = (U == 64) ? ~0ull : (uiw < upw ? (((1ull << U) - 1ull) << (uiw * U)) : 0ull);
    uint32_t w = 0u;
    if (valid) w = ctx->sl.hot[4 * ((size_t) hp.base[k] + (size_t) env * (size_t) S + (size_t) slot) + 3];  // (slot indices go up to 2^31: 64-bit word index)
    const bool empty = valid && (int) (w & 127u) <= 1;
    const uint64_t be = __ballot(empty) & unit_mask, be2 = __ballot(valid && (int) (w & 127u) <= 2) & unit_mask;
    if (unit_ok && slot == 0) {
        ctx->st.empt[(uint32_t) k * (uint32_t) hp.n_envs + (uint32_t) env] = (uint8_t) __popcll(be);
        // ... and what the slot pass of the step before this one would have left for the walk of the step behind this one (empt2)
        ctx->st.empt2[(sa.tick + 1u) & 1u][(uint32_t) k * (uint32_t) hp.n_envs + (uint32_t) env] = (uint8_t) __popcll(be2);
    }
}

// The walk of ONE env: station 0's draws, then station 1's, in the reference's consumption order (receive_car, CHS.hpp:1272-1316 /
// 1583-1627), leaving per unit flow / cars admitted / queue (StationArrays::fa) and per admitted car its three variates (SlotArrays::var)
// FORECOURT (k_compat_walk / k_env_walk): the env's forecourt draws of the step follow the stations' in the streams (hvs_step inside hy_step,
// HYD:250-260, behind both evs_step calls, MGR:149-180): the arrival level, then one mk_soc per arrival -- made here as well and left in
// EnvArrays::hv_pre for the tail, which then does not touch the streams at all
template <bool RESET, typename Stream, bool FORECOURT = false>
__device__ __forceinline__ void compat_walk_env(const DevCtx *__restrict__ ctx, const StepArgs &sa, const int env, Stream &rs) {
    const HubParams &hp = ctx->hp;
    const Tables &tb = ctx->tb;
    const StationArrays &st = ctx->st;
    const int64_t N = hp.n_envs;
    const uint32_t St = (uint32_t) (hp.S[0] + hp.S[1]);
    const uint32_t par = sa.tick & 1u;  // the step the draws belong to: its buffers
    const bool far = !RESET && sa.walk_far != 0;
    const bool cp = hp.constant_charging != 0;
    for (int k = 0; k < 2; k++) {
        const int S = hp.S[k];
        const bool fast = hp.type[k] == 0;
        const uint32_t sidx = (uint32_t) k * (uint32_t) N + (uint32_t) env;
        int line = 0, empties = S;
        if (far) {
            // two steps ahead of the slots (StepArgs::walk_far): the queue as the previous step's walk left it, and the slots that will be
            // empty = those with at most two slots of stay left when the step before that one ended, minus the cars the previous step's
            // walk admits, plus those of them that stay one slot at most
            const uint32_t wp = st.fa[par ^ 1u][sidx];
            line = (int) (wp >> 24);
            empties = (int) st.empt2[par][sidx] - (int) ((wp >> 16) & 255u) + (int) st.shrt[par ^ 1u][sidx];
        } else if (!RESET) {
            line = pkd_line(st.rec[4u * sidx + 3u]);
            empties = (int) st.empt[sidx];
        }
        const int mu = S / 2;  // round(charge_number / 2) on ints, CHS.hpp:1276
        int n_in;
        if (RESET) {
            const float cn = rs.normal_f((float) mu, 1.0f);
            int temp = (int) roundf(cn);
            temp = temp > mu + 3 ? mu + 3 : (temp < mu - 3 ? mu - 3 : temp);
            n_in = temp;
        } else {
            const int t_env = sa.env_clk ? clk_t(env_clk(sa, N, env)) : sa.t;  // per-env clocks: the env's own slot of day
            n_in = (int) tb.cnt[k][t_env * kLevels + rs.level()];
        }
        int tline = 0;
        for (int w = 0; w < line; w++) tline += (rs.level() >= (int) tb.thr_renege[w]) ? 1 : 0;
        int new_line = tline;
        int true_in = 0;
        for (int j = 0; j < n_in; j++) {
            const int m = new_line + j;
            const int thr = (int) tb.thr_balk[m < kBalkTab ? m : kBalkTab - 1];
            true_in += (rs.level() <= thr && j <= S) ? 1 : 0;
        }
        const int fl = fast ? n_in : true_in;
        const int as = (new_line + fl) < empties ? (new_line + fl) : empties;
        new_line = new_line + fl - as;
        new_line = new_line < kMaxLine ? new_line : kMaxLine;
        CHUB_G(u32x4) var = (CHUB_G(u32x4)) ctx->sl.var[par] + 2u * ((uint32_t) env * St + (uint32_t) (k ? hp.S[0] : 0));
        int n_short = 0;
        for (int rr = 0; rr < as; rr++) {  // ascending slot order == ascending rank
            const float soc = arrive_soc_from(rs.normal_d(7.0, 3.0));
            const int lev = rs.level();
            int late = (int) roundf(rs.normal_f(2.0f, 2.0f));  // mk_late_time("slow"), CHS.hpp:816-830
            late = late < 0 ? 0 : late;
            // add_car (CHS.hpp:864-877 / 1029-1042) HERE, where the car's variates are in registers: every lane of the walk that still has a car
            // to draw evaluates it, while in the slot pass the one lane in fifteen that admits a car made its whole wave pay for the f64 curve
            // work (round 6: 250 of a slot wave's 813 vector instructions).  Same functions on the same values: the same bits.
            const float target = uniform_level(lev, 80.0f, 100.0f);
            NewCar nc;
            if (fast) nc = make_car<0>(soc, lev, soc_to_time<0>(target, cp), late, cp);
            else nc = make_car<1>(soc, lev, soc_to_time<1>(target, cp), late, cp);
            var[2 * rr] = u32x4{__float_as_uint(nc.power), __float_as_uint(nc.t_target), __float_as_uint(nc.t_soc), (uint32_t) nc.stay | ((uint32_t) lev << 7)};
            var[2 * rr + 1] = u32x4{__float_as_uint(soc), 0u, 0u, 0u};
            // (what a walk two steps ahead needs of this step's admissions: how many of them stay one slot at most)
            n_short += (nc.stay <= 1) ? 1 : 0;
        }
        st.fa[par][sidx] = ((uint32_t) fl & 0xFFFFu) | ((uint32_t) (as > 0 ? as : 0) << 16) | ((uint32_t) new_line << 24);
        if (sa.walk_short) st.shrt[par][sidx] = (uint8_t) n_short;
    }
    if (FORECOURT && !RESET) {
        const int t_env = sa.env_clk ? clk_t(env_clk(sa, N, env)) : sa.t;
        CHUB_G(uint32_t) hv = ctx->ev.hv_pre[sa.tick & 1u] + (uint32_t) env * (uint32_t) hp.hv_w;
        const int arrive = (int) tb.cnt_hv[(uint32_t) t_env * (uint32_t) kLevels + (uint32_t) rs.level()];
        hv[0] = (uint32_t) arrive;
        for (int j = 0; j < arrive; j++) hv[1 + j] = __float_as_uint(arrive_soc_from(rs.normal_d(7.0, 3.0)));  // (arrive < hv_w: chub_create)
    }
}

// ---- the dense walk (round 6): the walk of 64 envs by ONE WAVE, in two kinds of phase.
// What is serial in a walk is the streams, and the streams are touched by cheap integer work only: the level draws, and the rejection loops of
// the polar normals (CompatStreamT::polar_d / polar_f).  What is expensive -- per new car two f64 logs, two divisions, two square roots and
// add_car's curve work, some 400 vector instructions -- depends on nothing but the accepted points.  With one env per lane all the way
// (compat_walk_env) a wave pays those 400 instructions once per car of its BUSIEST env (8 + 5 passes on an average step of the bench hub for
// 3 + 1.5 cars per env) on a chain that is the split step's long pole.  Here: phase A, lane = env, draws the raw points of up to kWalkCap cars
// into the wave's staging area in LDS (compacted: car j of the lanes that still have one, lane order); phase B, lane = CAR, evaluates them
// densely packed (3 + 2 passes) and writes the new cars where the slot pass looks for them.  Rounds repeat while any env has cars left.
// Same streams, same order of draws per env, the same functions on the same values: the same bits as compat_walk_env, which k_compat_small
// (a handful of envs on one wave's lanes) keeps.
constexpr int kWalkCap = 256;
constexpr int kWalkStageWords = kWalkCap * 7 + 64;  // per staged car: y, r2 of the SoC normal (f64), y, r2 of the extra-stay normal (f32), a word; + one counter per lane
struct WalkStage {
    double *yd, *r2d;
    float *yf, *r2f;
    uint32_t *meta;  // target level (10 bits) | owner lane << 10 | the car's admission rank << 16
    uint32_t *cnt;   // [64] per owner lane: cars of the round's station that stay one slot at most
};
__device__ __forceinline__ WalkStage walk_stage(uint32_t *base) {  // base: 8-byte aligned, kWalkStageWords words
    WalkStage g;
    g.yd = (double *) base;
    g.r2d = g.yd + kWalkCap;
    g.yf = (float *) (g.r2d + kWalkCap);
    g.r2f = g.yf + kWalkCap;
    g.meta = (uint32_t *) (g.r2f + kWalkCap);
    g.cnt = g.meta + kWalkCap;
    return g;
}
__device__ __forceinline__ void wave_sync_lds() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// KIND 0: a station's new cars (variates + add_car, written to SlotArrays::var / var_soc of env0 + owner lane); KIND 1: the forecourt's
// arrivals (one SoC normal each, written to EnvArrays::hv_pre).  `want`: how many the lane's env has to draw (0: none / no env).  All 64 lanes call.
template <int KIND, typename Stream>
__device__ __forceinline__ void walk_rounds(const DevCtx *__restrict__ ctx, const StepArgs &sa, Stream &rs, const WalkStage &sg, const int lane, const uint32_t env0,
                                            const int want, const int k, const bool fast) {
    const HubParams &hp = ctx->hp;
    const uint32_t St = (uint32_t) (hp.S[0] + hp.S[1]);
    const uint32_t par = sa.tick & 1u;
    const bool cp = hp.constant_charging != 0;
    int rr = 0;
    while (__any(want - rr > 0)) {
        const int rem = want - rr;
        // ---- phase A, lane = env: the raw points of the env's next cars -- as many per env as the staging area holds for all (the largest
        // quota Q with sum over envs of min(cars left, Q) <= kWalkCap), filed ENV BY ENV: lane l's cars at base_l .. base_l + q_l - 1, so
        // that phase B's neighbouring lanes hold one env's consecutive admission ranks and their records leave as one run of memory
        int Q = 0;
        {
            int total = 0;
            for (int j = 0;; j++) {
                const int c = __popcll(__ballot(rem > j));
                if (c == 0 || total + c > kWalkCap) break;  // (j = 0 always fits: c <= 64)
                total += c;
                Q = j + 1;
            }
        }
        const int mine = rem < Q ? (rem > 0 ? rem : 0) : Q;
        int base_l = mine;  // exclusive prefix sum of `mine` over the lanes (Hillis-Steele over the wave)
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int up = __shfl_up(base_l, d);
            if (lane >= d) base_l += up;
        }
        const int base = __shfl(base_l, 63);  // cars staged this round
        base_l -= mine;
        for (int j = 0; j < Q; j++) {
            if (j < mine) {
                const int i = base_l + j;
                double yd, r2d;
                rs.polar_d(yd, r2d);  // the SoC normal's accepted point (mk_soc, CHS.hpp:804-814 / HYD:259)
                sg.yd[i] = yd;
                sg.r2d[i] = r2d;
                uint32_t lev = 0u;
                if (KIND == 0) {
                    lev = (uint32_t) rs.level();  // target level (CHS.hpp:35-44)
                    float yf, r2f;
                    rs.polar_f(yf, r2f);          // mk_late_time's accepted point (CHS.hpp:816-830)
                    sg.yf[i] = yf;
                    sg.r2f[i] = r2f;
                }
                sg.meta[i] = lev | ((uint32_t) lane << 10) | ((uint32_t) (rr + j) << 16);
            }
        }
        wave_sync_lds();
        // ---- phase B, lane = car
        for (int i = lane; i < base; i += 64) {
            const uint32_t w = sg.meta[i];
            const uint32_t owner = (w >> 10) & 63u, rank = w >> 16;
            const float soc = arrive_soc_from(Stream::normal_of_polar_d(sg.yd[i], sg.r2d[i], 7.0, 3.0));
            if (KIND == 0) {
                const int lev = (int) (w & 1023u);
                int late = (int) roundf(Stream::normal_of_polar_f(sg.yf[i], sg.r2f[i], 2.0f, 2.0f));  // mk_late_time("slow")
                late = late < 0 ? 0 : late;
                // add_car (CHS.hpp:864-877 / 1029-1042)
                const float target = uniform_level(lev, 80.0f, 100.0f);
                NewCar nc;
                if (fast) nc = make_car<0>(soc, lev, soc_to_time<0>(target, cp), late, cp);
                else nc = make_car<1>(soc, lev, soc_to_time<1>(target, cp), late, cp);
                const uint32_t vi = (env0 + owner) * St + (uint32_t) (k ? hp.S[0] : 0) + rank;
                // ONE 32-byte record per new car (the hot record's four words + the arrival SoC): a sector written whole; neighbouring lanes hold
                // the same env's next admission ranks (the staging area is filled env by env), so a unit's records leave as one run
                ((CHUB_G(u32x4)) ctx->sl.var[par])[2u * vi] =
                    u32x4{__float_as_uint(nc.power), __float_as_uint(nc.t_target), __float_as_uint(nc.t_soc), (uint32_t) nc.stay | ((uint32_t) lev << 7)};
                ((CHUB_G(u32x4)) ctx->sl.var[par])[2u * vi + 1u] = u32x4{__float_as_uint(soc), 0u, 0u, 0u};
                // (what a walk two steps ahead needs of this step's admissions: how many of them stay one slot at most)
                if (nc.stay <= 1) atomicAdd(&sg.cnt[owner], 1u);
            } else {
                ctx->ev.hv_pre[par][(env0 + owner) * (uint32_t) hp.hv_w + 1u + rank] = __float_as_uint(soc);  // (rank < arrivals < hv_w: chub_create)
            }
        }
        wave_sync_lds();
        rr += mine;
    }
}

// The walk of 64 envs by one wave (lane = env in the serial phases): station 0's draws, then station 1's, then the forecourt's, in the
// reference's consumption order -- what compat_walk_env<.., FORECOURT = true> does for one env, with the new cars evaluated densely (above).
// live: the lane has an env to walk (inside the batch and named by the call).  All 64 lanes call.
template <bool RESET, typename Stream>
__device__ __forceinline__ void compat_walk_wave(const DevCtx *__restrict__ ctx, const StepArgs &sa, const uint32_t env0, const int lane, const bool live,
                                                 Stream &rs, uint32_t *stage_words) {
    const HubParams &hp = ctx->hp;
    const Tables &tb = ctx->tb;
    const StationArrays &st = ctx->st;
    const int64_t N = hp.n_envs;
    const uint32_t par = sa.tick & 1u;  // the step the draws belong to: its buffers
    const bool far = !RESET && sa.walk_far != 0;
    const int env = (int) env0 + lane;
    const WalkStage sg = walk_stage(stage_words);
    sg.cnt[lane] = 0u;
    wave_sync_lds();
    for (int k = 0; k < 2; k++) {
        const int S = hp.S[k];
        const bool fast = hp.type[k] == 0;
        const uint32_t sidx = (uint32_t) k * (uint32_t) N + (uint32_t) env;
        int as = 0, fl = 0, new_line = 0;
        if (live) {
            int line = 0, empties = S;
            if (far) {  // (two steps ahead of the slots: see compat_walk_env)
                const uint32_t wp = st.fa[par ^ 1u][sidx];
                line = (int) (wp >> 24);
                empties = (int) st.empt2[par][sidx] - (int) ((wp >> 16) & 255u) + (int) st.shrt[par ^ 1u][sidx];
            } else if (!RESET) {
                line = pkd_line(st.rec[4u * sidx + 3u]);
                empties = (int) st.empt[sidx];
            }
            const int mu = S / 2;  // round(charge_number / 2) on ints, CHS.hpp:1276
            int n_in;
            if (RESET) {
                const float cn = rs.normal_f((float) mu, 1.0f);
                int temp = (int) roundf(cn);
                temp = temp > mu + 3 ? mu + 3 : (temp < mu - 3 ? mu - 3 : temp);
                n_in = temp;
            } else {
                const int t_env = sa.env_clk ? clk_t(env_clk(sa, N, env)) : sa.t;  // per-env clocks: the env's own slot of day
                n_in = (int) tb.cnt[k][t_env * kLevels + rs.level()];
            }
            int tline = 0;
            for (int w = 0; w < line; w++) tline += (rs.level() >= (int) tb.thr_renege[w]) ? 1 : 0;
            new_line = tline;
            int true_in = 0;
            for (int j = 0; j < n_in; j++) {
                const int m = new_line + j;
                const int thr = (int) tb.thr_balk[m < kBalkTab ? m : kBalkTab - 1];
                true_in += (rs.level() <= thr && j <= S) ? 1 : 0;
            }
            fl = fast ? n_in : true_in;
            as = (new_line + fl) < empties ? (new_line + fl) : empties;
            new_line = new_line + fl - as;
            new_line = new_line < kMaxLine ? new_line : kMaxLine;
            as = as > 0 ? as : 0;
        }
        walk_rounds<0>(ctx, sa, rs, sg, lane, env0, as, k, fast);
        if (live) {
            st.fa[par][sidx] = ((uint32_t) fl & 0xFFFFu) | ((uint32_t) as << 16) | ((uint32_t) new_line << 24);
            if (sa.walk_short) st.shrt[par][sidx] = (uint8_t) sg.cnt[lane];
        }
        sg.cnt[lane] = 0u;  // (read and cleared by its own lane; the next station's phase B comes behind a wave barrier)
    }
    if (!RESET) {
        int arrive = 0;
        if (live) {
            const int t_env = sa.env_clk ? clk_t(env_clk(sa, N, env)) : sa.t;
            arrive = (int) tb.cnt_hv[(uint32_t) t_env * (uint32_t) kLevels + (uint32_t) rs.level()];
            ctx->ev.hv_pre[par][(uint32_t) env * (uint32_t) hp.hv_w] = (uint32_t) arrive;
        }
        walk_rounds<1>(ctx, sa, rs, sg, lane, env0, arrive, 0, false);
    }
}

// One workgroup's NW walks (256; k_slot_walk2: 64, one wave): the rings parked in LDS, walked, and the streams' state behind the draws
// written to the SHADOW of the step the draws belong to -- the stream buffer behind the committed one (CompatRng::g3 / minstd3: three buffers in
// rotation, StepArgs::rng_cur names the committed one): when the slot pass of that step has been launched the host moves rng_cur on, which IS
// the commit -- nothing is copied.  A walk two steps ahead (StepArgs::walk_far) starts from the previous step's shadow, rng_cur + 1, instead of the
// committed streams and writes rng_cur + 2.
template <bool RESET, int NW = 256, int LANES = NW>  // NW envs walked by LANES threads (LANES = 64 > NW = 32: half a wave's lanes walk, all of them evaluate the new cars)
__device__ __forceinline__ void compat_walk_block(const DevCtx *__restrict__ ctx, const StepArgs &sa, const uint32_t blk, uint32_t *s_ring, uint32_t *s_stage,
                                                  const uint32_t tid = threadIdx.x) {  // tid: the thread's number among the block's LANES; s_stage: LANES / 64 staging areas of kWalkStageWords
    static_assert(LANES >= NW && LANES % 64 == 0 && (NW % 64 == 0 || LANES == 64), "whole waves; a block of fewer than 64 envs is one wave");
    const int64_t N = ctx->hp.n_envs;
    const bool far = !RESET && sa.walk_far != 0;
    const uint32_t b_src = ((uint32_t) sa.rng_cur + (far ? 1u : 0u)) % 3u, b_dst = (b_src + 1u) % 3u;
    const uint32_t *g_src = (const uint32_t *) ctx->cr.g3[b_src];
    const uint32_t *m_src = (const uint32_t *) ctx->cr.minstd3[b_src];
    uint32_t *m_dst = (uint32_t *) ctx->cr.minstd3[b_dst];
    // the glibc rings of the workgroup's envs (128 bytes each, one contiguous run of memory) are parked in LDS for the walk: transposed, so
    // that the lanes of a wave hit different banks when each reads a word of its own ring
    const uint32_t env0 = blk * (uint32_t) NW;
    const uint32_t n_here = (uint32_t) N - env0 < (uint32_t) NW ? (uint32_t) N - env0 : (uint32_t) NW;
    {
        const u32x4 *src = (const u32x4 *) (g_src + (size_t) env0 * 32u);
        for (uint32_t j = tid; j < n_here * 8u; j += (uint32_t) LANES) {
            const u32x4 q = src[j];
            const uint32_t l = j >> 3, w = (j & 7u) << 2;
            s_ring[(w + 0u) * (uint32_t) NW + l] = q.x;
            s_ring[(w + 1u) * (uint32_t) NW + l] = q.y;
            s_ring[(w + 2u) * (uint32_t) NW + l] = q.z;
            s_ring[(w + 3u) * (uint32_t) NW + l] = q.w;
        }
    }
    __syncthreads();
    // (64 walks per wave: fewer -- 32 or 16 envs per wave, more waves -- measured slower: 115 / 129 vs 110 us per step at 65 536 envs)
    const int env = (int) (env0 + tid);
    const bool has_env = tid < (uint32_t) NW && env < (int) N;
    const bool live = has_env && in_group(sa, env);
    {
        CompatStreamT<RingLdsT<NW>> rs;
        rs.r.s = s_ring + (tid < (uint32_t) NW ? tid : 0u);
        rs.gf = rs.gr = 0u;
        rs.x = 1u;
        if (has_env) {
            rs.gf = rs.r.get(31);
            rs.gr = (rs.gf + 28u) % 31u;
            rs.x = m_src[env];  // (an env the call does not name: its streams as they are, in the shadow too)
        }
        compat_walk_wave<RESET>(ctx, sa, env0 + (tid & ~63u), (int) (tid & 63u), live, rs, s_stage + (tid >> 6) * (uint32_t) kWalkStageWords);
        if (has_env) {
            rs.r.set(31, rs.gf);
            m_dst[env] = rs.x;
        }
    }
    __syncthreads();
    {
        u32x4 *dst = (u32x4 *) (ctx->cr.g3[b_dst] + (size_t) env0 * 32u);
        for (uint32_t j = tid; j < n_here * 8u; j += (uint32_t) LANES) {
            const uint32_t l = j >> 3, w = (j & 7u) << 2;
            dst[j] = u32x4{s_ring[(w + 0u) * (uint32_t) NW + l], s_ring[(w + 1u) * (uint32_t) NW + l], s_ring[(w + 2u) * (uint32_t) NW + l],
                           s_ring[(w + 3u) * (uint32_t) NW + l]};
        }
    }
}

template <bool RESET>
__global__ __launch_bounds__(256) void k_compat_walk(const DevCtx *__restrict__ ctx, StepArgs sa) {
    __shared__ __attribute__((aligned(16))) uint32_t s_ring[32 * 256];
    __shared__ __attribute__((aligned(16))) uint32_t s_stage[4 * kWalkStageWords];
    compat_walk_block<RESET>(ctx, sa, blockIdx.x, s_ring, s_stage);
}

#ifndef CHUB_SPLIT2
#define CHUB_SPLIT2 1  // the split step's slot pass with two slots per lane (slot_body_split2); 0: slot_body_compat<.., SPLIT> for every step
#endif
template <int BLOCK>
__global__ __launch_bounds__(BLOCK, 7) void k_slot_split2(const DevCtx *__restrict__ ctx, StepArgs sa, int64_t nb0) {
    const HubParams &hp = ctx->hp;
    const int64_t bid = blockIdx.x;
    const int k = (bid >= nb0) ? 1 : 0;
    const int64_t bl = k ? bid - nb0 : bid;
    __shared__ float lds[(BLOCK / 64) * (3 * 128 + 16)];
    if (hp.type[k] == 0) slot_body_split2<0, BLOCK>(hp, sa, ctx->sl, ctx->st, ctx->cr, ctx->tb, k, bl, lds);
    else slot_body_split2<1, BLOCK>(hp, sa, ctx->sl, ctx->st, ctx->cr, ctx->tb, k, bl, lds);
}

// The slot pass of step i and the stream walks of step i + 1 in ONE launch: the walk -- one serial chain per env, a wave per SIMD for 20 to
// 40 us -- no longer stands between two slot passes but runs in the shadow of one.  It cannot look at what the slot pass beside it leaves,
// and does not need to: who leaves depends on the stays alone, so the slots that will be empty for step i + 1 are those with at most two
// slots of stay left after step i - 1 (StationArrays::empt2, left by that step's pass) minus the cars step i admits (its walk's word) plus
// those of them that stay one slot at most (StationArrays::shrt, from that walk: make_car's arithmetic); the queue is in that word too,
// and the streams continue from step i's shadow, which the slot pass beside it is committing (StepArgs::walk_far).  Workgroups
// [0, nwalk): 64 (or 32) walks each, on one wave (8 KB of rings + 7 KB of staging in LDS: a larger area would cost the slot workgroups their occupancy),
// at raised priority; the others: slot_body_split2.
// Envs per walk workgroup of k_slot_walk2 (one wave walks them): 64, or -- batches of at most kWalk2HalfMaxEnvs envs -- 32: twice the walking
// waves, each with half the cars to evaluate (all 64 lanes still evaluate).  A small batch's launch lasts as long as its longest walk
// (4096 envs: 27.0 -> 22.3 us, 16 384: 31.1 -> 26.8, 32 768: 37.7 -> 33.2, 40 000: 40.4 -> 39.7); a large one is bound by instruction issue,
// where the second set of serial phases costs more than the shorter chain gives (65 536 envs: 54.6 -> 59.1 us).
constexpr int64_t kWalk2HalfMaxEnvs = 40000;
#ifndef CHUB_WALK2_OCC
#define CHUB_WALK2_OCC 7  // workgroups per CU of k_slot_walk2 (sets its register budget: 72 VGPRs at 7, 80 at 6, 96 at 5)
#endif
template <int BLOCK, int kWalk2Envs>
__global__ __launch_bounds__(BLOCK, CHUB_WALK2_OCC) void k_slot_walk2(const DevCtx *__restrict__ ctx, StepArgs sa, StepArgs sw, int64_t nb0, int nwalk) {
    constexpr int kWords = (BLOCK / 64) * (3 * 128 + 16) > 32 * kWalk2Envs + kWalkStageWords ? (BLOCK / 64) * (3 * 128 + 16) : 32 * kWalk2Envs + kWalkStageWords;
    __shared__ __attribute__((aligned(16))) float lds[kWords];  // a walk workgroup: 8 KB of rings + the walking wave's staging area
    if ((int) blockIdx.x < nwalk) {
        // one wave of the workgroup walks, the others end at once (ended waves do not count at its barriers) -- the wave whose number is
        // the workgroup's modulo 4, so that the walks of the workgroups a CU receives do not all sit on the same SIMD
        if ((threadIdx.x >> 6) != (blockIdx.x & 3u)) return;
        __builtin_amdgcn_s_setprio(3);
        compat_walk_block<false, kWalk2Envs, 64>(ctx, sw, blockIdx.x, (uint32_t *) lds, (uint32_t *) lds + 32 * kWalk2Envs, threadIdx.x & 63u);
        return;
    }
    const HubParams &hp = ctx->hp;
    const int64_t bid = (int64_t) blockIdx.x - nwalk;
    const int k = (bid >= nb0) ? 1 : 0;
    const int64_t bl = k ? bid - nb0 : bid;
    if (hp.type[k] == 0) slot_body_split2<0, BLOCK>(hp, sa, ctx->sl, ctx->st, ctx->cr, ctx->tb, k, bl, lds);
    else slot_body_split2<1, BLOCK>(hp, sa, ctx->sl, ctx->st, ctx->cr, ctx->tb, k, bl, lds);
}

template <bool RESET, int BLOCK>
__global__ __launch_bounds__(BLOCK, 7) void k_slot_split(const DevCtx *__restrict__ ctx, StepArgs sa, int64_t nb0) {
    const HubParams &hp = ctx->hp;
    const int64_t bid = blockIdx.x;
    const int k = (bid >= nb0) ? 1 : 0;
    const int64_t bl = k ? bid - nb0 : bid;
    __shared__ float lds_f[BLOCK];  // (the scalar-load mode's rank pass: the only user of the scratch areas here)
    __shared__ uint32_t lds_u[2 * BLOCK];
    if (hp.type[k] == 0) slot_body_compat<0, RESET, BLOCK, true>(hp, sa, ctx->sl, ctx->st, ctx->cr, ctx->tb, k, bl, lds_f, lds_u);
    else slot_body_compat<1, RESET, BLOCK, true>(hp, sa, ctx->sl, ctx->st, ctx->cr, ctx->tb, k, bl, lds_f, lds_u);
}

// ----------------------------------------------------------------------------------------- k_env
// SAE J2601 target pressure (HYD:338-388 on the table HYD:232-233)
__device__ __forceinline__ double j2601_target(double p) {
    const double X[10] = {0.50, 5.00, 10.0, 15.0, 20.0, 30.0, 40.0, 50.0, 60.0, 70.0};
    const double Y[10] = {87.4, 81.0, 86.8, 86.1, 85.4, 83.8, 82.2, 80.4, 78.5, 76.1};
    if (p < X[1]) return Y[0];
    double r = p;
    bool hit = false;
#pragma unroll
    for (int a = 0; a < 8; a++) {
        const bool in = (a < 7) ? (X[1 + a] <= p && p < X[2 + a]) : (X[1 + a] <= p && p <= X[2 + a]);
        if (in && !hit) {
            r = (Y[2 + a] - Y[1 + a]) / (X[2 + a] - X[1 + a]) * (p - X[1 + a]) + Y[1 + a];
            hit = true;
        }
    }
    return r;
}
// x / c for a divisor known ahead of time, without the ~35-instruction division sequence on the tail's dependent chain:
// q = x * RN(1/c), one exact residual, one correction (Markstein; Brisebarre, Muller, Raina 2004).  The result is the
// correctly rounded quotient except for isolated (divisor-specific) x where it is one ulp off -- 1e-16 relative on f64
// values whose parity bar is 1e-5.
__device__ __forceinline__ double div_c(double x, double c, double rc) {
    const double q = x * rc;
    const double r = __fma_rn(-q, c, x);
    return __fma_rn(r, rc, q);
}
#define DIV_K(x, c) div_c((x), (double) (c), 1.0 / (double) (c))  // literal divisor: reciprocal folded at compile time

__device__ __forceinline__ double pressure_to_mass(double p) { return (6.3 * 1000) * DIV_K(p, 70); }  // HYD:390-391
// HYD:308-321
__device__ __forceinline__ void j2601_time_mass(double p0, double &time_need, double &mass_need) {
    const double target = j2601_target(p0);
    if (p0 < 5) time_need = DIV_K(69 - p0, 18.5) + (87.4 - 69) / 7.2;
    else time_need = (5 <= p0 && p0 < 70) ? DIV_K(target - p0, 18.5) : (target - p0) / 0.0;
    mass_need = pressure_to_mass(target) - pressure_to_mass(p0);
}

__device__ __forceinline__ void fcev_time_mass(float socf, double &tn, double &mn) {
    double soc = (double) socf;
    if (soc < 0.5) soc = 0.5;
    const double p0 = (soc * 0.01) * 70;  // _soc_to_pressure HYD:302-306
    j2601_time_mass(p0, tn, mn);
}

__device__ __forceinline__ double ou_sample(double &state, double theta, double sigma, double z) {  // REN:71-76
    const double dx = theta * (0.0 - state) + sigma * z;
    state += dx;
    return state;
}

#define CHUB_TEL(i, v)                                        \
    do {                                                      \
        if (tel_on) ev.telem[(size_t) (i) * (size_t) N + (size_t) env] = (v); \
    } while (0)

// The per-env tail of step() / reset(), lane = env.
constexpr int kEnvBlock = 256;  // compile-time (reading blockDim.x fetches the AQL packet)

// ---- the per-env tail of step() / reset() for one env per lane: the body of k_env (tables staged in LDS between the
// load burst and the arithmetic).
// What the tail of one env reads of its own state (PHILOX step).  k_step_fused requests it for the last wave's envs together with
// the wave's first slot loads, a whole slot pass ahead of its use.
struct TailIn {
    double ou_pv, ou_wd, ou_price, price_noise, cap;
    float a_el, a_fc;
    int pv_day, wd_day, q_len, hv_line;
    u32x4 drw;
    uint32_t drw_n;
};
__device__ __forceinline__ void tail_prefetch(TailIn &in, const TailArgs &ta, const uint32_t e32, const bool predrawn) {
    const uint32_t n32 = ta.n_envs;
    in.ou_pv = ta.ou[e32];
    in.ou_wd = ta.ou[n32 + e32];
    in.ou_price = ta.ou[2u * n32 + e32];
    in.price_noise = ta.price_noise[e32];
    if (ta.tail_act) {  // packed actions: the two tail actions have an array of their own
        typedef float f32x2_ __attribute__((ext_vector_type(2)));
        const f32x2_ tv = ((CHUB_G(const f32x2_)) ta.tail_act)[e32];
        in.a_el = tv.x;
        in.a_fc = tv.y;
    } else {
        const uint32_t ai = e32 * ta.act_dim + ta.s_tot;
        in.a_el = ta.actions[ai];
        in.a_fc = ta.actions[ai + 1u];
    }
    in.cap = ta.cap[e32];
    in.pv_day = ta.pv_day[e32];
    in.wd_day = ta.wd_day[e32];
    in.q_len = ta.q_len[e32];
    in.hv_line = ta.hv_line[e32];
    if (predrawn) {
        in.drw = ((CHUB_G(const u32x4)) ta.drw)[e32];
        in.drw_n = ta.drw_cnt[e32];
    }
}

// FUSED (k_step_fused): one WAVE runs the tails of up to 64 envs of its workgroup right behind their station records: the table rows
// are already in LDS (the caller's), the two records come from LDS (s_rec, [2 * local env + station]), `row` = the lane's place
// among the wave's output rows, env0 = the env of row 0, and the only synchronisation is the wave's own.
// The tail has two halves: what does not look at the station records (the slot's exogenous values, the FCEV forecourt, the exogenous
// update for the next slot) and what does (clamp, hydrogen step, netting, fuel cell, money, observation).  mid() is called between
// them by every lane: nothing in k_env; in k_step_fused the workgroup's third barrier and the record pass, so that the last wave's
// first half runs while the workgroup's first wave is still busy with the new cars.
struct NoMid {
    __device__ __forceinline__ void operator()() {}
    __device__ __forceinline__ void at1() {}  // (more places every lane with an env passes: k_steps_piped's tail wave meets the slot waves at some of them)
    __device__ __forceinline__ void at2() {}
    __device__ __forceinline__ void at3() {}
};
// EB: lanes of the workgroup that calls it (k_env: kEnvBlock; k_compat_small: its 512)
// TAPE (the parity instrument of include/chub.h, PHILOX handles): the tail's variates come from the caller -- the exogenous normals
// (sa.exo_z, f64 as the reference's numpy drew them), a reset's days (sa.exo_days), the forecourt's arrivals and their SoCs (sa.hv_tape) --
// and everything else is the production tail, instruction for instruction.
template <bool RESET, int MODE, bool MULTI, bool FUSED = false, typename Mid = NoMid, int EB = kEnvBlock, bool TAPE = false>
__device__ __forceinline__ void env_tail(const DevCtx *__restrict__ ctx, const StepArgs &sa, const int env, const bool live,
                                         const double *s_pv, const double *s_wd, const double *s_pv_now, const double *s_wd_now,
                                         const double *s_hy, const uint8_t *s_hv, float *s_out, const int env_block, const TailArgs &ta,
                                         const u32x4 *s_rec, const int local_env, const int env0_fused, const int rows_fused,
                                         const TailIn &pre, const bool use_pre, Mid &mid) {
    const HubParams &hp = ctx->hp;
    const EnvArrays &ev = ctx->ev;
    const CompatRng &cr = ctx->cr;
    const Tables &tb = ctx->tb;
    const int64_t N = hp.n_envs;  // (through the context pointer on purpose: round 5 measured the step 0.1 us SLOWER with this and hp.telemetry taken from
                                  // the kernel arguments -- the first touch of the context then comes later, in the middle of the dependent chain)
    // the clock of this lane's env: the launch's (lock-step), or its group's (per-env clocks: table rows then come straight
    // from global memory instead of the LDS copy of "the" slot of the day)
    constexpr bool multi = MULTI;  // its own instantiation: the lock-step kernel carries none of this
    const uint32_t my_clk = (multi && env < (int) N) ? env_clk(sa, N, env) : 0u;
    if (multi && env < (int) N)  // every env's clock moves to the other buffer: one step on, back to 0, or as it is
        sa.env_clk[(int64_t) ((sa.tick + 1u) & 1u) * N + env] = (uint16_t) (!live ? my_clk : (RESET ? 0u : clk_next(my_clk)));
    const int t_now = multi ? clk_t(my_clk) : sa.t;
    const int t_next = RESET ? 0 : (t_now + 1) % 96;
    const bool draw_price = multi ? ((my_clk >> 8) & 3u) == 0u : sa.draw_price != 0;
    const double price_last = multi ? tb.price[RESET ? 95 : t_now] : sa.price_last;
    const double price_prev = multi ? tb.price[(t_now + 95) % 96] : sa.price_prev;  // what the previous make_state saw as price[-1]
#define TAB_PV(d) (multi ? tb.pvT[t_next * 100 + (d)] : s_pv[d])
#define TAB_WD(d) (multi ? tb.wdT[t_next * 150 + (d)] : s_wd[d])
#define TAB_PV_NOW(d) (multi ? tb.pvT[t_now * 100 + (d)] : s_pv_now[d])
#define TAB_WD_NOW(d) (multi ? tb.wdT[t_now * 150 + (d)] : s_wd_now[d])
#define TAB_HY(i) (MODE == MODE_COMPAT ? hy_env[i] : s_hy[i])
#define TAB_HV(l) (multi ? tb.cnt_hv[(uint32_t) t_now * (uint32_t) kLevels + (uint32_t) (l)] : s_hv[l])
    // ---- prefetch: every per-env input is requested before the table staging and its barrier, and the PHILOX
    // words that do not depend on state (FCEV arrival level, the three OU normals) are drawn here too, so the
    // memory latencies of this latency-bound kernel overlap instead of queueing behind one another.
    const uint32_t e32 = (uint32_t) env, n32 = ta.n_envs;
    CHUB_STAMP_DECL(16);
    CHUB_STAMP_REAL(8);
    CHUB_STAMP(0);
#if CHUB_TRACE
    {
        uint32_t n_ = n32;
        asm volatile("" : "+s"(n_));
        CHUB_STAMP(1);  // kernel arguments here
    }
#endif
    // the table rows of this slot of the day (PV, wind, hy_table; COMPAT: FCEV counts) go out first, one element per lane, in
    // the same burst as the state loads: one memory round trip for everything (they are parked in LDS further down)
    double st_pv = 0.0, st_wd = 0.0, st_hy = 0.0, st_pv_now = 0.0, st_wd_now = 0.0;
    uint32_t st_hv = 0;
    const double sin_t = multi ? tb.sin96[t_next] : ta.sin_t;  // the observation's time feature
    const bool tel_on = hp.telemetry != 0;
    if (!FUSED) {
        const int i = threadIdx.x;
        if (!multi) {  // lock-step: the rows of the launch's slot of the day
            if (i < 100) st_pv = ta.pv_row[i];
            if (i < 150) st_wd = ta.wd_row[i];
            if (!RESET) {
                if (i < 100) st_pv_now = ta.pv_row_now[i];
                if (i < 150) st_wd_now = ta.wd_row_now[i];
            }
        }
        if (!RESET) {
            if (i < 102) st_hy = ta.hy_table[i];
            // (the split step's tail does not look the forecourt's arrivals up -- the walk did: no table row, and no hop through the context
            // pointer for its address in front of the per-env loads)
            if (MODE == MODE_COMPAT && !multi && !sa.hv_tape && i < kLevels / 4) st_hv = ((CHUB_G(const uint32_t)) tb.cnt_hv)[sa.t * (kLevels / 4) + i];
        }
    }
    // The device-side tick offset of graph replays is a scalar load from DEVICE memory (a miss in the scalar cache of every CU at the start of a
    // launch), and a wave cannot wait for its scalar loads one by one: issued here it sat in front of the per-env loads below, which wait for
    // kernel arguments (round 5: the compiler's s_waitcnt lgkmcnt(0) in front of the burst waited for it -- two round trips in a row).  The
    // steady-state PHILOX step needs the tick only for the forecourt's later arrivals, so it asks for the offset BEHIND the burst.
    constexpr bool tick_late = MODE == MODE_PHILOX && !RESET && !TAPE;
    const bool tick_now = MODE == MODE_PHILOX && (!tick_late || sa.fresh != 0);
    PhiloxCtx px{ta.key[0], ta.key[1], tick_now ? sa.tick + sload_u32(ta.tick_base, 0) : sa.tick, ta.gid0 + (uint32_t) env};
    double cap = 0.0, in_price_noise = 0.0;
    double ou_pv = 0.0, ou_wd = 0.0, ou_price = 0.0, z_pv = 0.0, z_wd = 0.0, z_pr = 0.0;
    float a_el_f = 0.0f, a_fc_f = 0.0f, P0f = 0.0f, P1f = 0.0f, mn0 = 0.0f, mx0 = 0.0f, mn1 = 0.0f, mx1 = 0.0f;
    int pv_day = 0, wd_day = 0, q_len = 0, hv_line = 0, F0i = 0, F1i = 0, ln0 = 0, ln1 = 0, hv_lev = 0, hv_arrive = 0;
    u32x4 drw_raw = {0u, 0u, 0u, 0u};
    uint32_t drw_n = 0u, hvw0 = 0u, hvw1 = 0u;
    const bool have_pre = FUSED && use_pre;  // the env's own state came in ahead of time (tail_prefetch)
    if (live) {
        if (have_pre) {
            ou_pv = pre.ou_pv; ou_wd = pre.ou_wd; ou_price = pre.ou_price; in_price_noise = pre.price_noise;
        } else {
            ou_pv = ta.ou[e32];
            ou_wd = ta.ou[n32 + e32];
            ou_price = ta.ou[2u * n32 + e32];
            in_price_noise = ta.price_noise[e32];
        }
        if (!FUSED) {  // (FUSED: the records are read from LDS behind mid(), where this workgroup writes them)
            const StationRec r0 = rec_load(ta.rec, e32), r1 = rec_load(ta.rec, n32 + e32);
            mn0 = r0.mn; P0f = r0.chg; mx0 = r0.mx; ln0 = pkd_line(r0.pkd); F0i = pkd_flow(r0.pkd);
            mn1 = r1.mn; P1f = r1.chg; mx1 = r1.mx; ln1 = pkd_line(r1.pkd); F1i = pkd_flow(r1.pkd);
        }
        if (!RESET && have_pre) {
            a_el_f = pre.a_el; a_fc_f = pre.a_fc; cap = pre.cap;
            pv_day = pre.pv_day; wd_day = pre.wd_day; q_len = pre.q_len; hv_line = pre.hv_line;
        } else if (!RESET) {
            if (ta.tail_act) {
                typedef float f32x2_ __attribute__((ext_vector_type(2)));
                const f32x2_ tv = ((CHUB_G(const f32x2_)) ta.tail_act)[e32];
                a_el_f = tv.x;
                a_fc_f = tv.y;
            } else {
                const uint32_t ai = e32 * ta.act_dim + ta.s_tot;
                a_el_f = ta.actions[ai];
                a_fc_f = ta.actions[ai + 1u];
            }
            cap = ta.cap[e32];
            pv_day = ta.pv_day[e32];
            wd_day = ta.wd_day[e32];
            q_len = ta.q_len[e32];
            hv_line = ta.hv_line[e32];
        }
        if (MODE == MODE_COMPAT && !RESET && sa.hv_tape) {  // the split step: the forecourt's draws wait where the walk left them -- the count and
            hvw0 = sa.hv_tape[e32 * (uint32_t) sa.hv_w];    // the first arrival's SoC with the burst, not as two round trips of their own behind it
            if (sa.hv_w > 1) hvw1 = sa.hv_tape[e32 * (uint32_t) sa.hv_w + 1u];
        }
        if (TAPE) {
            z_pv = sa.exo_z[e32 * 3u + 0u];
            z_wd = sa.exo_z[e32 * 3u + 1u];
            z_pr = sa.exo_z[e32 * 3u + 2u];
            if (!RESET) hv_arrive = (int) sa.hv_tape[e32 * (uint32_t) sa.hv_w];
        } else if (MODE == MODE_PHILOX && !RESET && !sa.fresh) {
            // this step's state-independent env draws (three OU normals, FCEV arrival count) were made one launch ahead by
            // the level blocks of k_env (draw_env_levels): 350 dependent instructions less on this latency-bound chain
            if (have_pre) {
                drw_raw = pre.drw;
                drw_n = pre.drw_n;
            } else {
                drw_raw = ((CHUB_G(const u32x4)) ta.drw)[e32];  // unpacked behind the table staging: no wait for it here
                drw_n = ta.drw_cnt[e32];
            }
        } else if (MODE == MODE_PHILOX) {
            const U4 ow = px.block(SITE_OU, 0, 0);  // word 0 pv, 1 wind, 2 price
            z_pv = (double) normal_from_word(tb.normal_icdf, tb.normal_tail, ow.v[0]);
            z_wd = (double) normal_from_word(tb.normal_icdf, tb.normal_tail, ow.v[1]);
            z_pr = (double) normal_from_word(tb.normal_icdf, tb.normal_tail, ow.v[2]);
            if (!RESET) {
                // the FCEV arrival level is state-independent: look its count up now, one byte straight from the table
                hv_lev = (int) (px.block(SITE_HV, 0, 0).v[0] % 1000u);
                hv_arrive = (int) tb.cnt_hv[(uint32_t) t_now * (uint32_t) kLevels + (uint32_t) hv_lev];
            }
        } else {
            z_pv = sa.exo_z[e32 * 3u + 0u];
            z_wd = sa.exo_z[e32 * 3u + 1u];
            z_pr = sa.exo_z[e32 * 3u + 2u];
        }
    }


    if (tick_late && !tick_now) {
        const uint32_t *tbp = ta.tick_base;
        asm volatile("" : "+s"(tbp) : : "memory");  // (not to be hoisted above the loads issued so far)
        px.tick += sload_u32(tbp, 0);
    }
    if (!FUSED) {
        // the table rows requested at the top arrive with the state loads; park them in LDS
        static_assert(EB >= 150, "one table element per lane");
        CHUB_STAMP(2);  // load burst issued
        const int i = threadIdx.x;
        if (i < 100) ((double *) s_pv)[i] = st_pv;
        if (i < 150) ((double *) s_wd)[i] = st_wd;
        if (!RESET) {
            if (i < 100) ((double *) s_pv_now)[i] = st_pv_now;
            if (i < 150) ((double *) s_wd_now)[i] = st_wd_now;
            if (i < 102) ((double *) s_hy)[i] = st_hy;
            if (MODE == MODE_COMPAT && i < kLevels / 4) ((uint32_t *) s_hv)[i] = st_hv;
        }
        __syncthreads();
    }
    CHUB_STAMP(3);  // rows parked, barrier passed: the loads have landed
    if (!TAPE && MODE == MODE_PHILOX && !RESET && !sa.fresh) {
        z_pv = (double) __uint_as_float(drw_raw.x);
        z_wd = (double) __uint_as_float(drw_raw.y);
        z_pr = (double) __uint_as_float(drw_raw.z);
        hv_arrive = (int) drw_n;
    }
    // stand-alone kernel: output rows go through LDS so that the workgroup writes its kEnvBlock consecutive rows (one
    // contiguous run of memory) with coalesced stores instead of 15 scattered 4-byte stores per lane
    const int row_w = sa.obs_stride;  // D (dense) or D + 2 (packed: obs, reward, done)
    auto flush_rows = [&]() {
        if (FUSED) {  // the wave's own rows: wave-level synchronisation only
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            float *dst = sa.obs + (size_t) env0_fused * (size_t) row_w;
            const int total = rows_fused * row_w;
            if ((((uintptr_t) dst) & 15u) == 0) {  // 16 bytes per lane and store: the wave's last dependent LDS round trips, a quarter as many
                typedef float f32x4 __attribute__((ext_vector_type(4)));
                const int quads = total >> 2;
                for (int i = (int) (threadIdx.x & 63u); i < quads; i += 64) ((f32x4 *) dst)[i] = ((const f32x4 *) s_out)[i];
                for (int i = (quads << 2) + (int) (threadIdx.x & 63u); i < total; i += 64) dst[i] = s_out[i];
            } else {
                for (int i = (int) (threadIdx.x & 63u); i < total; i += 64) dst[i] = s_out[i];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();  // s_out is reused by the wave's next group of envs
            return;
        }
        __syncthreads();
        const int env0 = env_block * EB;
        const int rows = (int) N - env0 < EB ? (int) N - env0 : EB;
        float *dst = sa.obs + (size_t) env0 * (size_t) row_w;
        const int total = rows * row_w;
        // per-env clocks: only the rows of the envs this launch serves (whole block served: the plain path)
        if (MULTI && sa.env_mask && !__syncthreads_and((live || env >= (int) N) ? 1 : 0)) {
            for (int i = threadIdx.x; i < total; i += EB)
                if (in_group(sa, env0 + i / row_w)) dst[i] = s_out[i];
        } else if ((((uintptr_t) dst) & 15u) == 0) {  // 16 bytes per lane and store: a quarter of the store instructions
            typedef float f32x4 __attribute__((ext_vector_type(4)));
            const int quads = total >> 2;
            for (int i = threadIdx.x; i < quads; i += EB) ((f32x4 *) dst)[i] = ((const f32x4 *) s_out)[i];
            for (int i = (quads << 2) + threadIdx.x; i < total; i += EB) dst[i] = s_out[i];
        } else {
            for (int i = threadIdx.x; i < total; i += EB) dst[i] = s_out[i];
        }
    };
    // what the first half hands to the second
    const double *hy_env = MODE == MODE_COMPAT ? (const double *) ev.hy_env + (size_t) e32 * 102u : nullptr;
    (void) hy_env;
    // COMPAT: the env's own hy_power_speed_list entry the electrolyser clamp looks at first (MGR:160-180; its index follows from the action
    // alone) is requested HERE, in front of the first half, instead of as a round trip of its own in the second
    double hy_req_pre = 0.0;
    if (MODE == MODE_COMPAT && !RESET && live) {
        int rq = (int) ceil((((double) a_el_f + 1) / 2) * 100);
        rq = rq < 0 ? 0 : (rq > 101 ? 101 : rq);
        hy_req_pre = hy_env[rq];
    }
    const double cap_mass = hp.cap_mass;
    double store_soc = 0.0, reward = 0.0, in_re_pv = 0.0, in_re_wd = 0.0, in_price_next = 0.0, total_mass_need = 0.0;
    double re_pv = 0.0, re_wd = 0.0, price_next = 0.0;
    int arrive = 0;
    uint32_t fold_n = 0;
    // ---- first half.  One pass of a do/while (here and below) so that lanes without an env skip the work but still reach mid()
    // and the workgroup barrier of flush_rows() at the same program point as everybody else
    do {
    if (!live) break;

    // COMPAT: the forecourt's draws either continue the env's streams here (one kernel per station, k_compat_small), or -- the split step --
    // were made by the walk behind the stations' and wait in sa.hv_tape (the tail then does not touch the streams)
    const bool hv_walked = MODE == MODE_COMPAT && sa.hv_tape != nullptr;
    CompatStream rs;
    if (MODE == MODE_COMPAT && !RESET && !hv_walked) rs.load(cr, sa.rng_cur, env);

    if (RESET) {
        // renew_reset (REN:51-53) + hy_reset (HYD:197-208)
        if (MODE == MODE_COMPAT || TAPE) {
            pv_day = sa.exo_days[e32 * 2u + 0u];
            wd_day = sa.exo_days[e32 * 2u + 1u];
        } else {
            const U4 o = px.block(SITE_DAY, 0, 0);
            pv_day = (int) (o.v[0] % 100u);
            wd_day = (int) (o.v[1] % 150u);
        }
        ev.pv_day[e32] = (int16_t) pv_day;
        ev.wd_day[e32] = (int16_t) wd_day;
        ev.q_len[e32] = 0;
        ev.hv_line[e32] = 0;
        cap = hp.init_soc * cap_mass;
        store_soc = hp.init_soc;
        CHUB_TEL(4, cap);
    } else {
        // What the previous make_state (MGR:344-361: the end of the previous step, or reset) produced for THIS slot is not kept as
        // state but re-derived, operation for operation, from what it was computed from: the table rows of this slot of the day,
        // the OU states as that make_state left them (the values loaded above), the price noise and the tariff it saw last.
        {
            double tp = TAB_PV_NOW(pv_day);
            if (tp > 0 && (pv_day % 2) == 0) tp += ou_pv * hp.renew_fluct1;  // REN:38-43
            in_re_pv = (tp > 0 ? tp : 0.0) * 5;
            double tw = TAB_WD_NOW(wd_day);
            tw += ou_wd * hp.renew_fluct1;                                    // REN:45-49
            in_re_wd = (tw > 0 ? tw : 0.0) * 1;
        }
        in_price_next = price_prev + in_price_noise;            // MGR:354-359
        // ---- hvs_step (HYD:253-285): FCEV arrivals -> J2601 -> 15-minute FIFO.  The reference's waiting list is unbounded.
        // Here: up to hp.qcap explicit entries, which is all a list that still gets served can hold (after a partial serve
        // it keeps at most arrive - 1 of the step's arrivals, HYD:270-279), plus -- once no prefix fits any more (nobody is
        // served again until reset) -- a folded tail: entry count and the running sums of time and mass in
        // the reference's left-to-right order, so that sum(needed_time_list) / sum(needed_hy_list) stay bit-identical.
        const int qcap = hp.qcap;
        double *qt = (double *) ev.q_time + (size_t) e32 * (size_t) qcap, *qm = (double *) ev.q_mass + (size_t) e32 * (size_t) qcap;
        if (MODE == MODE_COMPAT && !hv_walked) hv_lev = rs.level();
        arrive = MODE == MODE_COMPAT ? (hv_walked ? (int) hvw0 : (int) TAB_HV(hv_lev)) : hv_arrive;
        double total_mass = 0.0;
        const bool fcev_pre = !TAPE && MODE == MODE_PHILOX && !RESET && !sa.fresh;
        // the first arrival's SoC was drawn one launch ahead with the other env draws (level_block: same Philox counter); its
        // fueling time and mass follow from it alone
        double pre_tn = 0.0, pre_mn = 0.0;
        if (fcev_pre && arrive > 0) fcev_time_mass(__uint_as_float(drw_raw.w), pre_tn, pre_mn);
        bool stuck = (hv_line & 128) != 0;
        hv_line &= 127;
        if (fcev_pre && q_len == 0 && !stuck && arrive == 1 && pre_tn <= 15.0) {
            // the common case by far: an empty FIFO, one arrival, served within the slot (hvs_step leaves the FIFO empty and
            // the line at 0, HYD:281-283) -- nothing to read from or write to the queue arrays
            total_mass = pre_mn;
            if (hv_line != 0) {
                ev.hv_line[e32] = 0;
                hv_line = 0;
            }
        } else if (q_len > 0 || stuck || arrive > 0) {
            double total_time = 0.0;
            if (stuck) {  // q_len == 0: everything is folded
                total_time = ev.q_fold[2u * e32];
                total_mass = ev.q_fold[2u * e32 + 1u];
                fold_n = ev.q_fold_cnt[e32];
            }
            for (int i = 0; i < q_len; i++) {
                total_time += qt[i];
                total_mass += qm[i];
            }
            for (int j = 0; j < arrive; j++) {
                double tn, mn;
                if (fcev_pre && j == 0) {
                    tn = pre_tn;
                    mn = pre_mn;
                } else {
                    float socf;
                    if (hv_walked && j == 0) socf = __uint_as_float(hvw1);
                    else if (TAPE || hv_walked) socf = __uint_as_float(sa.hv_tape[e32 * (uint32_t) sa.hv_w + 1u + (uint32_t) j]);
                    else if (MODE == MODE_COMPAT) socf = arrive_soc_from(rs.normal_d(7.0, 3.0));
                    else socf = soc_from_word(tb.soc_d_icdf, px.block(SITE_HVSOC, (uint32_t) j, 0).v[0]);
                    fcev_time_mass(socf, tn, mn);
                }
                if (stuck) {
                    fold_n++;
                } else if (q_len < qcap) {  // always (qcap = 2 * max arrivals per step - 1, chub_create)
                    qt[q_len] = tn;
                    qm[q_len] = mn;
                    q_len++;
                }
                total_time += tn;
                total_mass += mn;
            }
            int hv_num = 0;
            if (total_time > 15.0) {
                bool found = false;
                if (!stuck) {
                    for (int i = 1; i <= arrive - 1; i++) {
                        int keep = q_len - i;
                        keep = keep < 0 ? 0 : keep;
                        double part = 0.0;
                        for (int j = 0; j < keep; j++) part += qt[j];
                        if (part <= 15.0) {
                            hv_line = i;
                            hv_num = keep;
                            found = true;
                            break;
                        }
                    }
                }
                if (found) {
                    for (int j = hv_num; j < q_len; j++) {
                        qt[j - hv_num] = qt[j];
                        qm[j - hv_num] = qm[j];
                    }
                    q_len -= hv_num;
                } else {
                    // no prefix fits: the list only grows from here on (every later prefix contains this one), so its entries
                    // are never looked at one by one again -- fold them
                    fold_n += (uint32_t) q_len;
                    q_len = 0;
                    stuck = true;
                    ev.q_fold[2u * e32] = total_time;
                    ev.q_fold[2u * e32 + 1u] = total_mass;
                    ev.q_fold_cnt[e32] = fold_n;
                }
            } else {
                hv_line = 0;
                q_len = 0;
            }
            ev.q_len[e32] = (uint8_t) q_len;
            ev.hv_line[e32] = (uint8_t) (hv_line | (stuck ? 128 : 0));
        } else if (hv_line != 0) {
            ev.hv_line[e32] = 0;  // empty FIFO, no arrivals: total time 0 <= 15 -> line = 0 (HYD:281-283)
            hv_line = 0;
        }
        total_mass_need = total_mass;
    }
    mid.at1();

    // ---- make_state (MGR:344-361), the exogenous update for the NEXT slot: it looks at neither records nor actions, so it sits in
    // the first half (same operations in the same order as when it followed the reward: every chain is its own)
    {
        double temp = TAB_PV(pv_day);
        if (temp > 0 && (pv_day % 2) == 0) {  // REN:38-43
            temp += ou_sample(ou_pv, .01, 1., z_pv) * hp.renew_fluct1;
            ev.ou[e32] = ou_pv;
        }
        re_pv = (temp > 0 ? temp : 0.0) * 5;
        temp = TAB_WD(wd_day);
        temp += ou_sample(ou_wd, .01, 1.5, z_wd) * hp.renew_fluct1;  // REN:45-49
        ev.ou[n32 + e32] = ou_wd;
        re_wd = (temp > 0 ? temp : 0.0) * 1;
        if (draw_price) {  // MGR:354-357
            price_next = ou_sample(ou_price, .1, 0.005, z_pr) * hp.price_fluct1;
            ev.ou[2u * n32 + e32] = ou_price;
            ev.price_noise[e32] = price_next;
            price_next += price_last;
        } else {
            price_next = price_last + in_price_noise;
        }
    }
    if (MODE == MODE_COMPAT && !RESET && !hv_walked) rs.store(cr, sa.rng_cur, env);
    } while (0);

    CHUB_STAMP(4);  // first half (exogenous values, forecourt, next slot's exogenous update) done
    mid();  // k_step_fused: the workgroup's third barrier + the station records (by this wave); nothing elsewhere
    if (FUSED && live) {  // the records this workgroup has just written, from LDS
        const u32x4 v0 = s_rec[2 * local_env], v1 = s_rec[2 * local_env + 1];
        mn0 = __uint_as_float(v0.x); P0f = __uint_as_float(v0.y); mx0 = __uint_as_float(v0.z); ln0 = pkd_line(v0.w); F0i = pkd_flow(v0.w);
        mn1 = __uint_as_float(v1.x); P1f = __uint_as_float(v1.y); mx1 = __uint_as_float(v1.z); ln1 = pkd_line(v1.w); F1i = pkd_flow(v1.w);
    }

    // ---- second half
    do {
    if (!live) break;
    if (!RESET) {
        const double a_el = ((double) a_el_f + 1) / 2;  // action_real[-1] <- action[-2]  (MGR:400-403)
        const double a_fc = ((double) a_fc_f + 1) / 2;  // action_real[-2] <- action[-1]  (MGR:395-398)
        const double P0 = (double) P0f, P1 = (double) P1f;
        const double F0 = (double) F0i, F1 = (double) F1i;
        double re_new_power = in_re_wd + in_re_pv;    // MGR:143
        const double charging_power = 0.0 + P0 + P1;  // MGR:157
        // ---- electrolyser request clamp against the grid limit (MGR:160-180)
        double hy_power_limit = 2000 + re_new_power - charging_power;
        hy_power_limit = hy_power_limit > 0 ? hy_power_limit : 0.0;
        double act_el = a_el;
        int req = (int) ceil(a_el * 100);
        req = req < 0 ? 0 : (req > 101 ? 101 : req);  // actions outside [-1,1] would index out of the table
        if ((MODE == MODE_COMPAT ? hy_req_pre : TAB_HY(req)) > hy_power_limit) {
            int ind = 0;
            while (ind < 102 && !(TAB_HY(ind) >= hy_power_limit)) ind++;
            // hy_power_speed_list_input[ind - 1]; python index -1 wraps to the last entry (1.0)
            act_el = (ind >= 102 || ind == 0) ? 0.01 * 100 : 0.01 * (ind - 1);
        }
        {
        // ---- hy_step (HYD:160-195): production clamp, electrolyser + compressor power, tank
        double must_chg = cap_mass * 0.1 - cap;
        must_chg = must_chg > 0 ? must_chg : 0.0;
        double upper_charge = cap_mass - cap;
        upper_charge = upper_charge > 0 ? upper_charge : 0.0;
        double charge_temp = act_el * hp.v_h_max * (15 * 60);
        charge_temp = charge_temp < upper_charge ? charge_temp : upper_charge;
        charge_temp = charge_temp > must_chg ? charge_temp : must_chg;
        double flow = DIV_K(charge_temp, 15 * 60);
        flow = flow < hp.v_h_max ? flow : hp.v_h_max;
        double ele_power = 0.0;
        if (hp.cells != 0.0) {  // Electrolyser.get_power, HYD:38-48
            const double v_H_mass = div_c(flow, hp.cells, hp.rc_cells);
            const double v_H_mol = DIV_K(v_H_mass, 2.02);
            const double v_H_L = v_H_mol * hp.v_M;
            const double v_H = v_H_L * 1000 * 60;
            const double temp = div_c(v_H * 2 * 96487, hp.v_M * 1000 * 60, hp.rc_vm60k);
            double power = temp * temp * 0.326 + temp * 1.476;
            ele_power = DIV_K(hp.cells * power, 1000);
        }
        const double cpr_power = DIV_K(DIV_K(DIV_K(flow, 2.02) * hp.cpr_w12, 0.8), 1000);  // Compressor.generate_W, HYD:74-82
        // sty_step (HYD:104-126)
        cap += flow * 15 * 60;
        double lower_change = cap - 0.1 * cap_mass;
        lower_change = lower_change > 0 ? lower_change : 0.0;
        const double hy_use = total_mass_need < lower_change ? total_mass_need : lower_change;
        const double not_meet = total_mass_need - hy_use;
        cap -= hy_use;
        cap -= cap * hp.hydro_loss;
        store_soc = div_c(cap, cap_mass, hp.rc_cap_mass);
        const double all_power_second = ele_power + cpr_power;
        const bool gen_hy = flow > 0.5;  // MGR:161,173-179
        mid.at2();
        // ---- renewable netting (MGR:183-213)
        double hydrogen_power = all_power_second;
        double ev_power_sum = charging_power;
        double e0 = P0, e1 = P1;
        // fc_rate = charging_power / (0 + P0 + P1) (MGR:187-191) divides a positive finite number by itself: exactly 1
        double used_renew = 0.0;
        if (re_new_power >= hydrogen_power) {
            re_new_power -= hydrogen_power;
            used_renew += hydrogen_power;
            hydrogen_power = 0.0;
            const double tmp = ev_power_sum;
            ev_power_sum = ev_power_sum - re_new_power;
            ev_power_sum = ev_power_sum > 0 ? ev_power_sum : 0.0;
            used_renew += tmp - ev_power_sum;
            const double sl_ = 0.0 + e0 + e1;
            if (sl_ > 0) {
                const double rate = ev_power_sum / sl_;
                e0 = rate * e0;
                e1 = rate * e1;
            }
        } else {
            hydrogen_power -= re_new_power;
            used_renew = re_new_power;
        }
        CHUB_TEL(11, e0);  // self.re_ev_power_list is taken before the fuel-cell rescale (MGR:212)
        CHUB_TEL(12, e1);
        // ---- fuel cell (MGR:215-227, HFC.use_cell HYD:409-430)
        double fc_power = gen_hy ? 0.0 : hp.fc_max_power * a_fc;
        if (fc_power > hp.fc_max_power) fc_power = hp.fc_max_power;
        else if (fc_power < 0) fc_power = 0.0;
        else if (fc_power > ev_power_sum) fc_power = ev_power_sum;
        double hy_to_use = DIV_K(fc_power * 1500, 119.6);
        hy_to_use = cap < hy_to_use ? cap : hy_to_use;
        fc_power = fc_power > 0 ? fc_power : 0.0;  // HYD:426 (not H2-limited)
        cap -= hy_to_use;                          // Store_SOC is not refreshed (HYD:428)
        ev_power_sum -= fc_power;
        {
            const double sl_ = 0.0 + e0 + e1;
            if (sl_ > 0) {
                const double rate_ = ev_power_sum / sl_;
                e0 = rate_ * e0;
                e1 = rate_ * e1;
            }
        }
        const double hy_loss = -6 / 1000.0 * hy_to_use;
        // ---- incomes and reward (MGR:233-269)
        const double real_price_dollar = in_price_next / 4;
        const double income_evs_fast = 0.42 / 4 * P0 - real_price_dollar * e0;
        const double income_evs_slow = 0.21 / 4 * P1 - real_price_dollar * e1;
        const double income_evs = income_evs_fast + income_evs_slow;
        const double income_evs_serve = 0.8 * (0.0 + F0 + F1);
        const double income_hys = 6 / 1000.0 * hy_use;
        const double not_meet_loss = -10 / 1000.0 * not_meet;
        const double hy_cost = -real_price_dollar * hydrogen_power;
        reward = DIV_K(income_hys + income_evs + income_evs_serve + hy_cost + 1 * hy_loss + not_meet_loss, 50);
        if (tel_on) {
            CHUB_TEL(0, act_el); CHUB_TEL(1, flow); CHUB_TEL(2, all_power_second); CHUB_TEL(4, cap);
            CHUB_TEL(5, total_mass_need); CHUB_TEL(6, hy_use); CHUB_TEL(7, not_meet); CHUB_TEL(8, fc_power);
            CHUB_TEL(9, hy_to_use); CHUB_TEL(10, used_renew); CHUB_TEL(13, hydrogen_power);
            CHUB_TEL(14, income_hys + income_evs + income_evs_serve + hy_cost); CHUB_TEL(15, reward);
            CHUB_TEL(19, (double) arrive); CHUB_TEL(20, (double) hv_line); CHUB_TEL(21, (double) q_len + (double) fold_n);
            // ev_power_list / ev_power_sum as the incomes and cumulated_draw_ele see them, after the fuel-cell rescale (MGR:219-224, 262)
            CHUB_TEL(24, e0); CHUB_TEL(25, e1); CHUB_TEL(26, ev_power_sum);
            CHUB_TEL(27, in_price_next);  // real_state[1] as this step found it: what real_price_dollar is a quarter of (MGR:234)
        }
        }
    }

    ev.cap[e32] = cap;

    // state_norm (MGR:318-342) of what make_state (MGR:362-373) collects, written straight to the output row
    float *obs = s_out + (FUSED ? (int) (threadIdx.x & 63u) : (int) threadIdx.x) * row_w;
    double *o64 = tel_on ? (double *) ev.obs64 + (size_t) e32 * hp.obs_dim : nullptr;
    int n = 0;
#define CHUB_OBS(v)                    \
    do {                               \
        const double v_ = (v);         \
        obs[n] = (float) v_;           \
        if (o64) o64[n] = v_;          \
        n++;                           \
    } while (0)
    CHUB_OBS(sin_t);
    CHUB_OBS(div_c(price_next - hp.price_mean, hp.price_std, hp.rc_price_std));
    if (hp.S[0] > 0) {
        const double half_range = (double) hp.transformer_limit[0] / 2, rc_hr = hp.rc_half_range[0];
        CHUB_OBS(div_c((double) mn0 - half_range, half_range, rc_hr));
        CHUB_OBS(div_c((double) P0f - half_range, half_range, rc_hr));
        CHUB_OBS(div_c((double) mx0 - half_range, half_range, rc_hr));
        CHUB_OBS(DIV_K((double) ln0, 5));
    }
    if (hp.S[1] > 0) {
        const double half_range = (double) hp.transformer_limit[1] / 2, rc_hr = hp.rc_half_range[1];
        CHUB_OBS(div_c((double) mn1 - half_range, half_range, rc_hr));
        CHUB_OBS(div_c((double) P1f - half_range, half_range, rc_hr));
        CHUB_OBS(div_c((double) mx1 - half_range, half_range, rc_hr));
        CHUB_OBS(DIV_K((double) ln1, 5));
    }
    CHUB_OBS(store_soc);
    CHUB_OBS(DIV_K(re_pv, 42 * 5));
    CHUB_OBS(DIV_K(re_wd, 92 * 1));
#undef CHUB_OBS
    if (!RESET) {
        const bool dn = (t_now + 1) >= 96;  // MGR:271-273
        if (sa.done_f32) {  // packed row: reward and done ride in the same LDS row
            obs[n] = (float) reward;
            obs[n + 1] = dn ? 1.0f : 0.0f;
        } else {
            sa.reward[(size_t) e32 * sa.reward_stride] = (float) reward;
            if (sa.done) sa.done[e32] = (uint8_t) (dn ? 1 : 0);
            if (sa.done_f32) sa.done_f32[(size_t) e32 * sa.reward_stride] = dn ? 1.0f : 0.0f;
        }
    }
    if (tel_on) {
        ev.reward64[e32] = reward;
        CHUB_TEL(3, store_soc); CHUB_TEL(16, re_pv); CHUB_TEL(17, re_wd); CHUB_TEL(18, price_next);
        CHUB_TEL(22, (double) pv_day); CHUB_TEL(23, (double) wd_day);
        // the station scalars make_state puts into real_state (MGR:364-368) and the arrivals income_evs_serve counts (MGR:246)
        CHUB_TEL(28, (double) mn0); CHUB_TEL(29, (double) P0f); CHUB_TEL(30, (double) mx0); CHUB_TEL(31, (double) ln0); CHUB_TEL(32, (double) F0i);
        CHUB_TEL(33, (double) mn1); CHUB_TEL(34, (double) P1f); CHUB_TEL(35, (double) mx1); CHUB_TEL(36, (double) ln1); CHUB_TEL(37, (double) F1i);
    }
    } while (0);
    CHUB_STAMP(5);  // second half (clamp, hydrogen step, netting, fuel cell, money, observation) done
    mid.at3();
    flush_rows();
#if CHUB_TRACE
    CHUB_STAMP(6);  // rows flushed
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    CHUB_STAMP(7);  // stores drained
    CHUB_STAMP_REAL(9);
    if (!FUSED && sa.stamps_env && threadIdx.x == 0) for (int i = 0; i < 10; i++) sa.stamps_env[(size_t) env_block * 16 + i] = stamp_[i];
#endif
#undef TAB_PV
#undef TAB_WD
#undef TAB_PV_NOW
#undef TAB_WD_NOW
#undef TAB_HY
#undef TAB_HV
}

// Next step's state-independent draws, lane u: [0, 2N) the station-level variates of unit u, [2N, 3N) the per-env draws
template <bool RESET, bool MULTI>
__device__ __forceinline__ void level_block(const DevCtx *__restrict__ ctx, const StepArgs &sa, const int64_t u, const int line_known = -1) {
    const HubParams &hp = ctx->hp;
    const int64_t N = hp.n_envs;
    if (u >= 3 * N) return;
    const int64_t env_ = u < N ? u : (u < 2 * N ? u - N : u - 2 * N);
    if (MULTI && !in_group(sa, env_)) return;
    // the slot of day the draws are for: the one after this launch's, on the env's own clock
    const int t_next = RESET ? 0 : ((MULTI ? clk_t(env_clk(sa, N, env_)) : sa.t) + 1) % 96;
    if (u < 2 * N) {
        const int kk = u >= N ? 1 : 0;
        // drawn AND decoded here, against the queue the slot kernel of this launch has just left in the unit's record
        const int line_now = line_known >= 0 ? line_known : pkd_line(ctx->st.rec[4u * (uint32_t) u + 3u]);
        ctx->st.pk[(sa.tick + 1u) & 1u][u] = draw_decoded_levels(hp, ctx->tb, CHUB_TICK(hp, sa.tick + 1u), t_next, kk, u - (int64_t) kk * N,
                                                                 line_now, hp.type[kk] == 0);
    } else {
        // the per-env draws (same Philox sites and counters the tail would use itself)
        const uint32_t e = (uint32_t) (u - 2 * N);
        const Tables &tb = ctx->tb;
        PhiloxCtx px{hp.key[0], hp.key[1], CHUB_TICK(hp, sa.tick + 1u), (uint32_t) (hp.env_id0 + e)};
        const U4 ow = px.block(SITE_OU, 0, 0);  // word 0 pv, 1 wind, 2 price
        const uint32_t hv_lev = px.block(SITE_HV, 0, 0).v[0] % 1000u;
        u32x4 d;
        d.x = __float_as_uint(normal_from_word(tb.normal_icdf, tb.normal_tail, ow.v[0]));
        d.y = __float_as_uint(normal_from_word(tb.normal_icdf, tb.normal_tail, ow.v[1]));
        d.z = __float_as_uint(normal_from_word(tb.normal_icdf, tb.normal_tail, ow.v[2]));
        const uint32_t cnt = (uint32_t) tb.cnt_hv[(uint32_t) t_next * (uint32_t) kLevels + hv_lev];
        d.w = cnt != 0u ? __float_as_uint(soc_from_word(tb.soc_d_icdf, px.block(SITE_HVSOC, 0, 0).v[0])) : 0u;  // the first arrival's SoC
        ((CHUB_G(u32x4)) ctx->ev.drw[(sa.tick + 1u) & 1u])[e] = d;
        ctx->ev.drw_cnt[(sa.tick + 1u) & 1u][e] = (uint8_t) cnt;
    }
}

template <bool RESET, int MODE, bool MULTI, bool TAPE = false>
__global__ __launch_bounds__(kEnvBlock) void k_env(const DevCtx *__restrict__ ctx, StepArgs sa, TailArgs ta, int nb_env) {
    __shared__ double s_pv[100], s_wd[150], s_pv_now[100], s_wd_now[150], s_hy[102];
    __shared__ __attribute__((aligned(16))) uint8_t s_hv[kLevels];
    __shared__ __attribute__((aligned(16))) float s_out[kEnvBlock * 16];  // output rows: obs_dim + 2 <= 15 floats
    if (MODE == MODE_PHILOX && (int) blockIdx.x >= nb_env) {
        // the last blocks of the grid: next step's station-level draws, one lane per (station, env).
        // They are pure VALU work and fill the issue slots the latency-bound tail waves leave empty.
        int seg = 0;
        int64_t env_ = 0;
#if CHUB_TRACE
        CHUB_STAMP_DECL(16);
        CHUB_STAMP_REAL(8);
#endif
        // XCD-aware order here too where the launch divides evenly (every XCD's eighth of the envs a whole number of workgroups, the tail
        // workgroups a multiple of 8): level workgroup j runs on XCD j % 8 and takes that XCD's envs -- station 0's units, station 1's,
        // then the env draws -- so that its CUs touch the same eighth of every array as the tail workgroups beside them
        const uint32_t jb = blockIdx.x - (uint32_t) nb_env, n32 = (uint32_t) ctx->hp.n_envs, per = n32 >> 3;  // envs per XCD
        bool have;
        if (!MULTI && ta.xcd && (n32 & 2047u) == 0u && ((uint32_t) nb_env & 7u) == 0u) {
            const uint32_t x = jb & 7u, i = jb >> 3, wpx = per / kEnvBlock;  // i-th of the 3 * wpx workgroups of XCD x
            seg = (int) (i / wpx);
            env_ = (int64_t) (x * per + (i - (uint32_t) seg * wpx) * kEnvBlock + threadIdx.x);
            have = seg < 3;
        } else {
            have = range_unit(sa, (int64_t) jb * kEnvBlock + threadIdx.x, 3, seg, env_);
        }
        if (have) level_block<RESET, MULTI>(ctx, sa, (int64_t) seg * ctx->hp.n_envs + env_);
#if CHUB_TRACE
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        CHUB_STAMP_REAL(9);
        if (sa.stamps_env && threadIdx.x == 0) {  // (the level workgroups' rows follow the tail workgroups': entry and exit on the shared clock)
            sa.stamps_env[(size_t) blockIdx.x * 16 + 8] = stamp_[8];
            sa.stamps_env[(size_t) blockIdx.x * 16 + 9] = stamp_[9];
        }
#endif
        return;
    }
    const int blk = (int) xcd_order(blockIdx.x, (uint32_t) nb_env, MULTI ? 0u : ta.xcd) + (MULTI ? sa.env_lo / kEnvBlock : 0);  // a call on a subset: the blocks of its range of envs only
    const int env = blk * kEnvBlock + (int) threadIdx.x;
    TailIn none;  // (the stand-alone tail loads its inputs itself)
    NoMid nomid;
    env_tail<RESET, MODE, MULTI, false, NoMid, kEnvBlock, TAPE>(ctx, sa, env, env < (int) ta.n_envs && (!MULTI || in_group(sa, env)), s_pv, s_wd, s_pv_now,
                                                                s_wd_now, s_hy, s_hv, s_out, blk, ta, nullptr, 0, 0, 0, none, false, nomid);
}

// ---------------------------------------------------------------------------------------- k_env_walk: COMPAT, the split step's second launch
// The tails of step i and the stream walks of step i + 1 in ONE launch (lock-step steps of every env): both are one wave per SIMD --
// the tail a latency chain that waits two thirds of its time, the walk bound by instruction issue -- and they do not depend on each other:
// the walk needs the queue lengths and empty-slot counts the slot pass of step i has just left, and reads the streams as that slot pass
// committed them (the tail does not touch them: its forecourt draws were made by step i's walk); it writes the streams' state behind its
// draws to the shadow, so if a reset comes instead of step i + 1 nothing has happened.  Workgroups [0, nb_env): env_tail; the others: walks.
__global__ __launch_bounds__(256) void k_env_walk(const DevCtx *__restrict__ ctx, StepArgs sa, TailArgs ta, StepArgs sw, int nb_env) {
    __shared__ __attribute__((aligned(16))) uint32_t lds[32 * 256 + 4 * kWalkStageWords];  // the walk's rings (32 KB) + its waves' staging areas / the tail's table rows and output rows (22 KB)
    if ((int) blockIdx.x >= nb_env) {
        __builtin_amdgcn_s_setprio(3);  // the walks are the launch's long pole: the tail wave on the same SIMD takes the issue slots they leave
        compat_walk_block<false>(ctx, sw, blockIdx.x - (uint32_t) nb_env, lds, lds + 32 * 256);
        return;
    }
    double *s_pv = (double *) lds, *s_wd = s_pv + 100, *s_pv_now = s_wd + 150, *s_wd_now = s_pv_now + 100, *s_hy = s_wd_now + 150;  // 602 doubles
    uint8_t *s_hv = (uint8_t *) (s_hy + 102 + 2);                                                                                   // kLevels bytes, 16-byte aligned
    float *s_out = (float *) (s_hv + 1008);                                                                                          // kEnvBlock * 16 floats
    static_assert((602 + 2) * 8 + 1008 + kEnvBlock * 16 * 4 <= 32 * 256 * 4 && kLevels <= 1008, "the tail's LDS inside the walk's");
    const int env = (int) blockIdx.x * kEnvBlock + (int) threadIdx.x;
    TailIn none;
    NoMid nomid;
    env_tail<false, MODE_COMPAT, false>(ctx, sa, env, env < (int) ta.n_envs, s_pv, s_wd, s_pv_now, s_wd_now, s_hy, s_hv, s_out, (int) blockIdx.x, ta, nullptr, 0, 0,
                                        0, none, false, nomid);
}

// ---------------------------------------------------------------------------------------- k_compat_small: COMPAT, a handful of envs, ONE launch
// The reference-exact mode is what the drop-in class runs, one env at a time (EvcsspManagerEnv_v6: test/env_test.py's loop), and there
// a step is nothing but launch latency and the serial walk of the env's two reference streams.  When every env of the handle fits ONE
// workgroup, the split step (k_compat_walk -> k_slot_split -> k_env) runs inside one launch of kCompatSmallBlock lanes, its parts as
// ROLES of the workgroup's waves:
//   wave 0             lane = env: the walk (compat_walk_env: station 0's draws, then station 1's), later the tail
//   waves 1 .. 3       station 0's units, waves 4 .. 7 station 1's (slot_body_compat<.., SPLIT>): loads, on / off, car_step along the
//                      curves and the departures run WHILE wave 0 walks; the bodies stop at the workgroup barrier in front of their
//                      first look at what the walk came to
// then the admissions, sums and records, another barrier, and env_tail (its forecourt draws continue the env's streams where the walk
// left them).  What crosses waves goes through memory as between the three launches (release / acquire at agent scope around the
// workgroup barriers).  Same functions, same order of draws: the results are those of the three launches.
static_assert(64 * (1 + kCompatSmallWaves0 + kCompatSmallWaves1) == kCompatSmallBlock, "wave roles of k_compat_small");
struct SmallMid {
    __device__ __forceinline__ void operator()() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
};
template <bool RESET>
__global__ __launch_bounds__(kCompatSmallBlock) void k_compat_small(const DevCtx *__restrict__ ctx, StepArgs sa, TailArgs ta) {
    constexpr int BLOCK = kCompatSmallBlock;
    __shared__ float lds_f[BLOCK];
    __shared__ uint32_t lds_u[2 * BLOCK];
    __shared__ double s_pv[100], s_wd[150], s_pv_now[100], s_wd_now[150], s_hy[102];
    __shared__ __attribute__((aligned(16))) uint8_t s_hv[kLevels];
    __shared__ __attribute__((aligned(16))) float s_out[kEnvBlock * 16];
    const HubParams &hp = ctx->hp;
    const int wave = (int) (threadIdx.x >> 6);
    const int k = wave <= kCompatSmallWaves0 ? 0 : 1, wave0 = k ? 1 + kCompatSmallWaves0 : 1;
    SmallMid mid;
    if (!RESET && sa.empt_fresh) {
        // the units' empty-slot counts are not the previous pass's (the first step after create / chub_set_state): counted here
        if (wave >= 1) {
            const int lane = (int) (threadIdx.x & 63u), U = hp.U[k], S = hp.S[k];
            const int upw = 64 / U, uiw = lane / U, slot = lane - uiw * U;
            const int env = (wave - wave0) * upw + uiw;
            const bool unit_ok = uiw < upw && env < (int) hp.n_envs;
            const bool valid = unit_ok && slot < S;
            const uint64_t unit_mask = (U == 64) ? ~0ull : (uiw < upw ? (((1ull << U) - 1ull) << (uiw * U)) : 0ull);
            uint32_t w = 0u;
            if (valid) w = ctx->sl.hot[4 * ((size_t) hp.base[k] + (size_t) env * (size_t) S + (size_t) slot) + 3];
            const uint64_t be = __ballot(valid && (int) (w & 127u) <= 1) & unit_mask;
            if (unit_ok && slot == 0) ctx->st.empt[(uint32_t) k * (uint32_t) hp.n_envs + (uint32_t) env] = (uint8_t) __popcll(be);
        }
        mid();
    }
    if (wave == 0) {
        const int env = (int) threadIdx.x;
        if (env < (int) hp.n_envs) {
            CompatStream rs;
            rs.load(ctx->cr, sa.rng_cur, env);
            compat_walk_env<RESET>(ctx, sa, env, rs);
            rs.store(ctx->cr, sa.rng_cur, env);
        }
        mid();  // (the barrier the slot waves reach inside their bodies)
    } else if (hp.type[k] == 0) {
        slot_body_compat<0, RESET, BLOCK, true, SmallMid>(hp, sa, ctx->sl, ctx->st, ctx->cr, ctx->tb, k, 0, lds_f, lds_u, wave0, mid);
    } else {
        slot_body_compat<1, RESET, BLOCK, true, SmallMid>(hp, sa, ctx->sl, ctx->st, ctx->cr, ctx->tb, k, 0, lds_f, lds_u, wave0, mid);
    }
    mid();  // the station records are in
    const int env = (int) threadIdx.x;
    TailIn none;
    NoMid nomid;
    env_tail<RESET, MODE_COMPAT, false, false, NoMid, BLOCK>(ctx, sa, env, env < (int) ta.n_envs, s_pv, s_wd, s_pv_now, s_wd_now, s_hy, s_hv, s_out, 0, ta,
                                                                    nullptr, 0, 0, 0, none, false, nomid);
}

// ---------------------------------------------------------------------------------------- k_step_fused: the whole step in ONE launch
// Small batches (C2: 4096 envs; every shard of an 8-GPU strong-scaling run): both step kernels are launch- and latency-bound
// there -- a handful of waves per CU, 4-5 us each of which 1.7 us is the kernel boundary -- and the chip has room for every
// workgroup at once, which is what made the fused form lose at 65 536 envs (lingering tail waves taking wave slots from the
// memory-bound slot work).  So: the packed slot body as it is, and behind the station records the workgroup goes on by itself --
//   last wave:  the per-env tails of the workgroup's envs (env_tail, records from LDS, table rows prefetched with the slot loads)
//   wave 0:     next step's station-level draws of the workgroup's units, decoded against the queue lengths just computed
//   wave 1:     next step's per-env draws
// Same functions, same Philox counters, same operations as k_slot_packed + k_env: bit-identical results
// (tests: chub_options.fused_step forced on / off).  PHILOX, lock-step, stations of up to 64 piles.
// The next step's draws of a one-launch workgroup (k_step_fused / k_step_tailwave / k_steps_fused / k_steps_piped): role 1 = the station-level draws of
// its units, decoded against the queue lengths the admission pass left in s_unit; role 2 = its envs' draws; any other role: nothing.
__device__ __forceinline__ void draw_next_step(const int role, const DevCtx *__restrict__ ctx, const StepArgs &sa, const PackedArgs &pa, const uint32_t *s_unit) {
    const int lane = threadIdx.x & 63;
    const int epb = (int) pa.epb, N = (int) pa.n_envs, env_first = (int) blockIdx.x * epb;
    if (role == 1) {
        for (int i = lane; i < 2 * epb; i += 64) {
            const int e = i >> 1, k = i & 1, env = env_first + e;
            if (env >= N) continue;
            int line;
            if ((k ? pa.S[1] : pa.S[0]) == 0u) {  // a station without piles: its queue moves on as in the body's record pass
                const int want = dk_want(pa.pk[(uint32_t) (k ? N : 0) + (uint32_t) env]);
                line = want < kMaxLine ? want : kMaxLine;
            } else {
                line = pkd_line(s_unit[i]);
            }
            level_block<false, false>(ctx, sa, (int64_t) k * N + env, line);
        }
    } else if (role == 2) {
        for (int i = lane; i < epb; i += 64)
            if (env_first + i < N) level_block<false, false>(ctx, sa, 2 * (int64_t) N + env_first + i);
    }
}
// The table rows a tail WAVE (k_step_tailwave, k_steps_piped) keeps in LDS for its envs: requested with the per-env state, parked behind the
// wave's first barrier (one wave: two or three elements per lane)
struct TailWaveRows {
    double r_pv[2], r_pvn[2], r_wd[3], r_wdn[3], r_hy[2];
    __device__ __forceinline__ void request(const TailArgs &ta, const bool with_hy) {
        const int lane = threadIdx.x & 63;
#pragma unroll
        for (int q = 0; q < 2; q++) {
            const int i = lane + 64 * q;
            r_pv[q] = r_pvn[q] = r_hy[q] = 0.0;
            if (i < 100) {
                r_pv[q] = ta.pv_row[i];
                r_pvn[q] = ta.pv_row_now[i];
            }
            if (i < 102 && with_hy) r_hy[q] = ta.hy_table[i];
        }
#pragma unroll
        for (int q = 0; q < 3; q++) {
            const int i = lane + 64 * q;
            r_wd[q] = r_wdn[q] = 0.0;
            if (i < 150) {
                r_wd[q] = ta.wd_row[i];
                r_wdn[q] = ta.wd_row_now[i];
            }
        }
    }
    __device__ __forceinline__ void park(double *s_pv, double *s_wd, double *s_pv_now, double *s_wd_now, double *s_hy, const bool with_hy) const {
        const int lane = threadIdx.x & 63;
#pragma unroll
        for (int q = 0; q < 2; q++) {
            const int i = lane + 64 * q;
            if (i < 100) {
                s_pv[i] = r_pv[q];
                s_pv_now[i] = r_pvn[q];
            }
            if (i < 102 && with_hy) s_hy[i] = r_hy[q];
        }
#pragma unroll
        for (int q = 0; q < 3; q++) {
            const int i = lane + 64 * q;
            if (i < 150) {
                s_wd[i] = r_wd[q];
                s_wd_now[i] = r_wdn[q];
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // (the wave's own LDS rows, read back by other lanes of it)
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
};
struct TailPrefetch {
    const TailArgs &ta;
    double *s_pv, *s_wd, *s_pv_now, *s_wd_now, *s_hy;
    double r_pv, r_wd, r_pv_now, r_wd_now, r_hy;
    TailIn tin;          // the last wave: the state of its first 64 envs
    int tail_env;        // ... of this env (-1: none)
    bool predrawn;
    __device__ __forceinline__ void prefetch() {
        const int i = threadIdx.x;
        if (tail_env >= 0) tail_prefetch(tin, ta, (uint32_t) tail_env, predrawn);
        r_pv = r_wd = r_pv_now = r_wd_now = r_hy = 0.0;
        if (i < 100) {
            r_pv = ta.pv_row[i];
            r_pv_now = ta.pv_row_now[i];
        }
        if (i < 150) {
            r_wd = ta.wd_row[i];
            r_wd_now = ta.wd_row_now[i];
        }
        if (i < 102) r_hy = ta.hy_table[i];
    }
    __device__ __forceinline__ void park() {
        const int i = threadIdx.x;
        if (i < 100) {
            s_pv[i] = r_pv;
            s_pv_now[i] = r_pv_now;
        }
        if (i < 150) {
            s_wd[i] = r_wd;
            s_wd_now[i] = r_wd_now;
        }
        if (i < 102) s_hy[i] = r_hy;
    }
};

template <int BLOCK, int T, bool BITS, bool TAPE>
__device__ __forceinline__ void step_fused_body(const DevCtx *__restrict__ ctx, const StepArgs &sa, const PackedArgs &pa_in, const TailArgs &ta) {
    static_assert(BLOCK >= 150 && BLOCK / 64 >= 3, "one table element per lane; three waves with work of their own");
    __shared__ uint32_t q_new[BLOCK * T];
    __shared__ uint32_t q_cnt[2];
    __shared__ uint64_t s_ball[BLOCK * T / 64 + 2];
    __shared__ __attribute__((aligned(16))) int s_acc[2 * BLOCK * T * kAccCopies];
    __shared__ uint32_t s_unit[BLOCK * T / 2];
    __shared__ uint32_t s_uinfo[BLOCK * T / 2];
    __shared__ __attribute__((aligned(16))) u32x4 s_rec[BLOCK * T / 2];
    __shared__ double s_pv[100], s_wd[150], s_pv_now[100], s_wd_now[150], s_hy[102];
    __shared__ __attribute__((aligned(16))) float s_out[64 * 16];
    PackedArgs pa = pa_in;
    asm volatile("" : "+s"(pa.S[0]), "+s"(pa.S[1]), "+s"(pa.type[0]), "+s"(pa.type[1]), "+s"(pa.n_envs), "+s"(pa.epb), "+s"(pa.magic),
                      "+s"(pa.cls_delta), "+s"(pa.state), "+s"(pa.rec), "+s"(pa.pk), "+s"(pa.actions), "+s"(pa.cls0), "+s"(pa.ttab2));
    TailPrefetch hook{ta, s_pv, s_wd, s_pv_now, s_wd_now, s_hy, 0.0, 0.0, 0.0, 0.0, 0.0, TailIn(), -1, sa.fresh == 0 && !TAPE};
    {   // the lanes of the last wave that will run a tail: which env's state to request up front
        const int le = (int) (threadIdx.x & 63u), env = (int) blockIdx.x * (int) pa_in.epb + le;
        if ((int) (threadIdx.x >> 6) == BLOCK / 64 - 1 && le < (int) pa_in.epb && env < (int) pa_in.n_envs) hook.tail_env = env;
    }
    const int role = slot_body_packed<BLOCK, T, TAPE, false, false, false, true, TailPrefetch, BITS>(ctx->hp, sa, pa, ctx->tb, blockIdx.x, q_cnt, q_new,
                                                                                                      s_ball + 1, s_acc, s_unit, hook, s_rec, s_uinfo);
    constexpr int WAVES = BLOCK / 64;
    const int lane = threadIdx.x & 63;
    const int epb = (int) pa.epb, N = (int) pa.n_envs, env_first = (int) blockIdx.x * epb;
    // The body has returned in front of the workgroup's third barrier (behind it every new car is in its slot and in the LDS sums).
    // The last wave runs the first half of its first 64 envs' tails before it goes there -- the half that needs no station record --
    // while the first wave is still busy with the new cars' Philox blocks; then the barrier, the records (by the last wave), the
    // second half.  The other waves pass the barrier and go on to next step's draws.
    struct RecordsMid {
        const PackedArgs &pa;
        int *s_acc;
        const uint32_t *s_unit;
        u32x4 *s_rec;
        bool first;  // the first group of envs: the barrier and the records are still ahead
        __device__ __forceinline__ void at1() {}
        __device__ __forceinline__ void at2() {}
        __device__ __forceinline__ void at3() {}
        __device__ __forceinline__ void operator()() {
            if (!first) return;
            __syncthreads();
            packed_records<BLOCK, T, false, false, false, true, BITS>(pa, blockIdx.x, s_acc, s_unit, s_rec);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // s_rec was written by other lanes of this wave
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    };
    if (role != WAVES) __syncthreads();
    if (role == 1 || role == 2) {
        draw_next_step(role, ctx, sa, pa, s_unit);
    } else if (role == WAVES) {
        for (int c = 0; c < epb; c += 64) {  // the workgroup's envs, 64 at a time (the wave's own LDS traffic: wave-level synchronisation)
            const int le = c + lane, env = env_first + le;
            const bool live = le < epb && env < N;
            int rows = epb - c < 64 ? epb - c : 64;
            rows = N - (env_first + c) < rows ? N - (env_first + c) : rows;
            RecordsMid mid{pa, s_acc, s_unit, s_rec, c == 0};
            env_tail<false, MODE_PHILOX, false, true, RecordsMid, kEnvBlock, TAPE>(ctx, sa, env, live, s_pv, s_wd, s_pv_now, s_wd_now, s_hy, nullptr, s_out, 0, ta,
                                                                                   s_rec, live ? le : 0, env_first + c, rows > 0 ? rows : 0, hook.tin, c == 0, mid);
        }
    }
}

template <int BLOCK, int T, bool BITS = false, bool TAPE = false>
__global__ __launch_bounds__(BLOCK, 4) void k_step_fused(const DevCtx *__restrict__ ctx, StepArgs sa, PackedArgs pa_in, TailArgs ta) {
    step_fused_body<BLOCK, T, BITS, TAPE>(ctx, sa, pa_in, ta);
}

// ---- The one-launch step with the tails on a wave of their own (k_step_tailwave; hubs of 8 piles and more: at most 64 envs per workgroup).
// What k_steps_piped's stamps showed holds inside ONE step too: the tail's first half (the slot's exogenous values, the forecourt, the next
// slot's exogenous update: ~1 us) looks at nothing the slot phases produce, the next step's draws (~2 us of Philox blocks and dependent table
// reads on two waves) wait for the admission's queue lengths only, and only the second half of the tail needs the station records.  So a fifth
// wave starts the tails with the kernel, beside the slot phases; slot waves 0 / 1 pass the fourth barrier right behind the third and make
// the next step's draws beside everything that follows; slot waves 2 / 3 serve the new cars and end; the tail wave turns the LDS sums into the
// station records behind the fourth barrier and runs the second half.  Chain: slot phases + new cars + records + second half, instead of slot
// phases + new cars + draws | first half + records + second half.  Same functions, same Philox counters: bit-identical (tests: every test of
// the one-launch step runs through it; chub_options.fused_step forced on / off).
template <int BLOCK, int T, bool BITS>
struct TailWaveMid {
    const PackedArgs &pa;
    int *s_acc;
    const uint32_t *s_unit;
    u32x4 *s_rec;
    __device__ __forceinline__ void at1() { __syncthreads(); }  // #2
    __device__ __forceinline__ void at2() {}
    __device__ __forceinline__ void at3() {}
    __device__ __forceinline__ void operator()() {
        __syncthreads();  // #3
        __syncthreads();  // #4: every new car is in its slot and in the LDS sums
        packed_records<BLOCK, T, false, false, false, true, BITS>(pa, blockIdx.x, s_acc, s_unit, s_rec);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // s_rec was written by other lanes of this wave
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
};
template <int BLOCK, int T, bool BITS = false>
__global__ __launch_bounds__(BLOCK + 64, 2) void k_step_tailwave(const DevCtx *__restrict__ ctx, StepArgs sa, PackedArgs pa_in, TailArgs ta) {
    constexpr int WAVES = BLOCK / 64;
    static_assert(WAVES == 4, "two waves draw while two serve the new cars");
    __shared__ uint32_t q_new[BLOCK * T];
    __shared__ uint32_t q_cnt[2];
    __shared__ uint64_t s_ball[BLOCK * T / 64 + 2];
    __shared__ __attribute__((aligned(16))) int s_acc[2 * BLOCK * T * kAccCopies];
    __shared__ uint32_t s_unit[BLOCK * T / 2];
    __shared__ uint32_t s_uinfo[BLOCK * T / 2];
    __shared__ __attribute__((aligned(16))) u32x4 s_rec[BLOCK * T / 2];
    __shared__ double s_pv[100], s_wd[150], s_pv_now[100], s_wd_now[150], s_hy[102];
    __shared__ __attribute__((aligned(16))) float s_out[64 * 16];
    PackedArgs pa = pa_in;
    asm volatile("" : "+s"(pa.S[0]), "+s"(pa.S[1]), "+s"(pa.type[0]), "+s"(pa.type[1]), "+s"(pa.n_envs), "+s"(pa.epb), "+s"(pa.magic),
                      "+s"(pa.cls_delta), "+s"(pa.state), "+s"(pa.rec), "+s"(pa.pk), "+s"(pa.actions), "+s"(pa.cls0), "+s"(pa.ttab2));
    const int wave = __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    const int epb = (int) pa.epb, N = (int) pa.n_envs, env_first = (int) blockIdx.x * epb;
    if (wave < WAVES) {
        NoHook hook;
        const int role = slot_body_packed<BLOCK, T, false, false, false, false, true, NoHook, BITS, true>(ctx->hp, sa, pa, ctx->tb, blockIdx.x, q_cnt, q_new,
                                                                                                            s_ball + 1, s_acc, s_unit, hook, nullptr, s_uinfo);
        __syncthreads();  // #4 (waves 0 / 1: right behind the third; waves 2 / 3: behind their new cars)
        draw_next_step(role, ctx, sa, pa, s_unit);
        return;
    }
    // ---- the tail wave
    const int env = env_first + lane;
    const bool live = lane < epb && env < N;
    TailIn tin = TailIn();
    if (live) tail_prefetch(tin, ta, (uint32_t) env, sa.fresh == 0);
    TailWaveRows tr;
    tr.request(ta, true);
    __syncthreads();  // #1
    tr.park(s_pv, s_wd, s_pv_now, s_wd_now, s_hy, true);
    int rows = epb < 64 ? epb : 64;
    rows = N - env_first < rows ? N - env_first : rows;
    TailWaveMid<BLOCK, T, BITS> mid{pa, s_acc, s_unit, s_rec};  // #2 (at1: behind the forecourt), #3 and #4 + the records (between the halves)
    env_tail<false, MODE_PHILOX, false, true, TailWaveMid<BLOCK, T, BITS>, kEnvBlock, false>(ctx, sa, env, live, s_pv, s_wd, s_pv_now, s_wd_now, s_hy, nullptr, s_out, 0, ta,
                                                                                       s_rec, live ? lane : 0, env_first, rows > 0 ? rows : 0, tin, true, mid);
}

// ---- A SPAN of lock-step steps in ONE launch ("multi-step persistent kernel for the random-policy bench").
// k_step_fused gives every workgroup whole envs -- their slots, their station records, their tails, the next step's draws -- and nothing
// crosses workgroups (MGR:136-302: the reference's envs are separate objects), so a workgroup can go from one step to the next BY ITSELF:
// no grid-wide synchronisation, one workgroup barrier between steps (every store of a step is then out and visible to the workgroup's own
// waves, which share the CU's vector cache).  What a step needs beyond the previous step's state arrives as today: the step's action rows
// (chub_run_steps's batches, step i reads batch i % n_batches) and its packed output block (step i writes block i & 1) -- so the policy is
// open loop over the span (the random-policy bench, replayed action tapes), and of the span's outputs the last two blocks remain.  Every
// per-step kernel argument is a function of the step number (clock, Philox tick, tariff, table rows, buffer parities): same functions,
// same Philox counters, bit-identical to the steps issued one by one.  What it saves is the kernel boundary between two steps of a
// launch-bound batch (the one-launch step is boundary + five dependent round trips + an f64 chain).
struct SpanArgs {
    int32_t n_steps, pc0;         // steps in the span (it ends at a day boundary at the latest); price_count before its first step (MGR:354)
    uint32_t first;               // index of its first step
    int32_t n_batches, D;
    const float *actions[8];      // chub_run_steps's action batches (n_batches <= 8)
    float *packed[2];             // ... and its two packed output blocks
    CHUB_G(const uint32_t) pk[2];     // StationArrays::pk, EnvArrays::drw / drw_cnt by tick parity
    CHUB_G(const uint32_t) drw[2];
    CHUB_G(const uint8_t) drw_cnt[2];
    CHUB_G(const double) price;   // Tables::price, pvT, wdT, sin96
    CHUB_G(const double) pvT;
    CHUB_G(const double) wdT;
    CHUB_G(const double) sin96;
};
template <int BLOCK, int T>
__global__ __launch_bounds__(BLOCK, 2) void k_steps_fused(const DevCtx *__restrict__ ctx, StepArgs sa0, PackedArgs pa0, TailArgs ta0, SpanArgs sp) {
    for (int s = 0; s < sp.n_steps; s++) {
        if (s) __syncthreads();
        StepArgs sa = sa0;
        PackedArgs pa = pa0;
        TailArgs ta = ta0;
        const uint32_t i = sp.first + (uint32_t) s;
        const int t = sa0.t + s, t_next = (t + 1) % 96;
        sa.t = t;
        sa.tick = sa0.tick + (uint32_t) s;
        sa.draw_price = ((sp.pc0 + s) % 4 == 0) ? 1 : 0;
        sa.price_last = sp.price[t];               // AGG:147
        sa.price_prev = sp.price[(t + 95) % 96];
        const uint32_t b = i % (uint32_t) sp.n_batches;
        const float *act = sp.actions[0];
#pragma unroll
        for (int j = 1; j < 8; j++) act = b == (uint32_t) j ? sp.actions[j] : act;
        float *out = (i & 1u) ? sp.packed[1] : sp.packed[0];
        sa.actions = act;
        sa.obs = out;
        sa.reward = out + sp.D;
        sa.done_f32 = out + sp.D + 1;
        if (s) sa.fresh = 0;  // (the previous step of this launch left this step's draws)
        const bool odd = (sa.tick & 1u) != 0u;
        pa.pk = odd ? sp.pk[1] : sp.pk[0];
        pa.actions = (CHUB_G(const float)) act;
        pa.tick = sa.tick;
        ta.drw = odd ? sp.drw[1] : sp.drw[0];
        ta.drw_cnt = odd ? sp.drw_cnt[1] : sp.drw_cnt[0];
        ta.actions = (CHUB_G(const float)) act;
        ta.pv_row = sp.pvT + t_next * 100;
        ta.wd_row = sp.wdT + t_next * 150;
        ta.pv_row_now = sp.pvT + t * 100;
        ta.wd_row_now = sp.wdT + t * 150;
        ta.sin_t = sp.sin96[t_next];
        step_fused_body<BLOCK, T, false, false>(ctx, sa, pa, ta);
    }
}

// ---- The same span with the TAILS A STEP BEHIND, on a wave of their own (k_steps_piped; hubs of 8 piles and more: at most 64 envs per
// workgroup).  In k_steps_fused a step is three chains end to end -- the slot phases up to the new cars (three dependent round trips, three
// barriers: ~2.4 us), the next step's draws (Philox blocks and dependent table reads on two waves: ~1.7 us) and, on the last wave alone, the
// station records and the tails (~600 dependent f64 operations, a wave64 f64 operation every 8 cycles: ~3.5 us) -- and none of them waits for
// the next one's RESULT: the slot phases of step s + 1 read slot words, decoded draws and action rows, none of which the tails of step s write
// (level_block takes the queue length from the admission pass, not from the tail), and the draws need the queue lengths the admission left,
// not the new cars.  So:
//   slot waves 0, 1:  behind the third barrier the next step's draws (units / envs) WHILE
//   slot waves 2, 3:  serve the new cars;  fourth barrier, next step
//   a fifth wave:     the station records and the tails of step s - 1 while the slot waves run step s: the LDS sums and queue words of a step
//                     live in two buffers by step parity, the table rows and output rows in areas nobody else touches; after the slot waves'
//                     last step it runs the last tails alone (a wave that has ended no longer counts at the barrier).
// gfx950 has ONE barrier per workgroup, so the tail wave takes part in the slot waves' four barriers per step: behind its load requests, behind
// the records, behind the forecourt (Mid::at1) and in front of its output flush (Mid::at3) -- four pieces of the tail against the four phases of
// the slot step, placed by the stamps of tools/experiments/piped_stamps.py so that neither side waits long for the other.
// Same functions, same Philox counters, same operations per env: bit-identical to steps issued one by one (tests).
#ifdef CHUB_PIPED_STAMPS  // (measurement build: when the tail wave of workgroup 0 reaches and leaves each of a step's four barriers)
__device__ unsigned long long g_piped_stamps[16];
#define PIPED_STAMP(k) do { if (stamp_on) g_piped_stamps[k] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define PIPED_STAMP(k) do { } while (0)
#endif
struct PipedMid {
    bool stamp_on;
    __device__ __forceinline__ void at1() { PIPED_STAMP(4); __syncthreads(); PIPED_STAMP(5); }
    __device__ __forceinline__ void at2() {}
    __device__ __forceinline__ void at3() { PIPED_STAMP(6); __syncthreads(); PIPED_STAMP(7); }
    __device__ __forceinline__ void operator()() {}
};
__device__ __forceinline__ void span_args(const StepArgs &sa0, const PackedArgs &pa0, const TailArgs &ta0, const SpanArgs &sp, const int s,
                                          StepArgs &sa, PackedArgs &pa, TailArgs &ta) {
    sa = sa0;
    pa = pa0;
    ta = ta0;
    const uint32_t i = sp.first + (uint32_t) s;
    const int t = sa0.t + s, t_next = (t + 1) % 96;
    sa.t = t;
    sa.tick = sa0.tick + (uint32_t) s;
    sa.draw_price = ((sp.pc0 + s) % 4 == 0) ? 1 : 0;
    sa.price_last = sp.price[t];               // AGG:147
    sa.price_prev = sp.price[(t + 95) % 96];
    const uint32_t b = i % (uint32_t) sp.n_batches;
    const float *act = sp.actions[0];
#pragma unroll
    for (int j = 1; j < 8; j++) act = b == (uint32_t) j ? sp.actions[j] : act;
    float *out = (i & 1u) ? sp.packed[1] : sp.packed[0];
    sa.actions = act;
    sa.obs = out;
    sa.reward = out + sp.D;
    sa.done_f32 = out + sp.D + 1;
    if (s) sa.fresh = 0;  // (the previous step of this launch left this step's draws)
    const bool odd = (sa.tick & 1u) != 0u;
    pa.pk = odd ? sp.pk[1] : sp.pk[0];
    pa.actions = (CHUB_G(const float)) act;
    pa.tick = sa.tick;
    ta.drw = odd ? sp.drw[1] : sp.drw[0];
    ta.drw_cnt = odd ? sp.drw_cnt[1] : sp.drw_cnt[0];
    ta.actions = (CHUB_G(const float)) act;
    ta.pv_row = sp.pvT + t_next * 100;
    ta.wd_row = sp.wdT + t_next * 150;
    ta.pv_row_now = sp.pvT + t * 100;
    ta.wd_row_now = sp.wdT + t * 150;
    ta.sin_t = sp.sin96[t_next];
}
template <int BLOCK, int T>
__global__ __launch_bounds__(BLOCK + 64, 2) void k_steps_piped(const DevCtx *__restrict__ ctx, StepArgs sa0, PackedArgs pa0, TailArgs ta0, SpanArgs sp) {
    constexpr int WAVES = BLOCK / 64;
    static_assert(WAVES == 4, "two waves draw while two serve the new cars");
    __shared__ uint32_t q_new[BLOCK * T];
    __shared__ uint32_t q_cnt[2];
    __shared__ uint64_t s_ball[BLOCK * T / 64 + 2];
    __shared__ __attribute__((aligned(16))) int s_acc2[2][2 * BLOCK * T * kAccCopies];  // a step's LDS sums and queue words, by step parity: the tail wave
    __shared__ uint32_t s_unit2[2][BLOCK * T / 2];                                      // turns them into station records while the next step is under way
    __shared__ uint32_t s_uinfo[BLOCK * T / 2];
    __shared__ __attribute__((aligned(16))) u32x4 s_rec[BLOCK * T / 2];                 // the tail wave's: the records ...
    __shared__ double s_pv[100], s_wd[150], s_pv_now[100], s_wd_now[150], s_hy[102];    // ... the table rows
    __shared__ __attribute__((aligned(16))) float s_out[64 * 16];                       // ... and its output rows
    const int wave = __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6));
    const int n = sp.n_steps;
    if (wave < WAVES) {
        // ---- the slot waves: step s = the packed slot body up to its third barrier (#1 #2 #3); then waves 0 / 1 the next step's draws, waves 2 / 3
        // the new cars; the fourth barrier (#4); the next step
        for (int s = 0; s < n; s++) {
            StepArgs sa;
            PackedArgs pa;
            TailArgs ta;
            span_args(sa0, pa0, ta0, sp, s, sa, pa, ta);
            asm volatile("" : "+s"(pa.S[0]), "+s"(pa.S[1]), "+s"(pa.type[0]), "+s"(pa.type[1]), "+s"(pa.n_envs), "+s"(pa.epb), "+s"(pa.magic),
                              "+s"(pa.cls_delta), "+s"(pa.state), "+s"(pa.rec), "+s"(pa.pk), "+s"(pa.actions), "+s"(pa.cls0), "+s"(pa.ttab2));
            NoHook hook;
            int *const s_acc = s_acc2[s & 1];
            uint32_t *const s_unit = s_unit2[s & 1];
            const int role = slot_body_packed<BLOCK, T, false, false, false, false, true, NoHook, false, true>(ctx->hp, sa, pa, ctx->tb, blockIdx.x, q_cnt, q_new,
                                                                                                                s_ball + 1, s_acc, s_unit, hook, nullptr, s_uinfo);
            draw_next_step(role, ctx, sa, pa, s_unit);  // (waves 0 / 1; the queue lengths are the admission's: in s_unit since the third barrier)
            __syncthreads();  // #4
        }
        return;  // (a wave that has ended no longer counts at the workgroup's barrier: the tail wave runs the last step's records and tails by itself)
    }
    // ---- the tail wave: during step s the station records and the tails of step s - 1
    for (int s = 0; s <= n; s++) {
        if (s == 0) {  // nothing to do during the first step: its four barriers
            __syncthreads();
            __syncthreads();
            __syncthreads();
            __syncthreads();
            continue;
        }
        StepArgs sa;
        PackedArgs pa;
        TailArgs ta;
        span_args(sa0, pa0, ta0, sp, s - 1, sa, pa, ta);
        const int lane = threadIdx.x & 63;
        const int epb = (int) pa.epb, N = (int) pa.n_envs, env_first = (int) blockIdx.x * epb;
        const int env = env_first + lane;
        const bool live = lane < epb && env < N;
        TailIn tin = TailIn();
        if (live) tail_prefetch(tin, ta, (uint32_t) env, sa.fresh == 0);
        TailWaveRows tr;
        tr.request(ta, s == 1);  // (the hydrogen table does not change: parked once)
        const bool stamp_on = blockIdx.x == 0 && (threadIdx.x & 63) == 0 && s == n - 2;
        (void) stamp_on;
        PIPED_STAMP(0);
        __syncthreads();  // #1
        PIPED_STAMP(1);
        // the station records of step s - 1 (every car's sums are in since that step's fourth barrier), into the record array and into s_rec
        packed_records<BLOCK, T, false, false, false, true, false>(pa, blockIdx.x, s_acc2[(s - 1) & 1], s_unit2[(s - 1) & 1], s_rec);
        PIPED_STAMP(2);
        __syncthreads();  // #2
        PIPED_STAMP(3);
        tr.park(s_pv, s_wd, s_pv_now, s_wd_now, s_hy, s == 1);  // (its wave-level fence also covers the records above: read back by other lanes of this wave)
        int rows = epb < 64 ? epb : 64;
        rows = N - env_first < rows ? N - env_first : rows;
        PipedMid mid{stamp_on};  // #3 (at1: behind the forecourt), #4 (at3: in front of the flush)
        env_tail<false, MODE_PHILOX, false, true, PipedMid, kEnvBlock, false>(ctx, sa, env, live, s_pv, s_wd, s_pv_now, s_wd_now, s_hy, nullptr, s_out, 0, ta,
                                                                              s_rec, live ? lane : 0, env_first, rows > 0 ? rows : 0, tin, true, mid);
        PIPED_STAMP(8);  // (flushed)
    }
}
#ifdef CHUB_PIPED_STAMPS
extern "C" int chub_debug_piped_stamps(unsigned long long *out) {
    return (int) hipMemcpyFromSymbol(out, HIP_SYMBOL(g_piped_stamps), sizeof(unsigned long long) * 16);
}
#endif

// COMPAT only: HySystem.__init__ (HYD:154-158) builds hy_power_speed_list with 101 REAL hy_step()s from the initial tank:
// each draws the FCEV arrival level and one mk_soc per arrival from the env's two streams (HYD:250-259), serves the
// 15-minute FIFO, clamps the production against the tank and moves the tank; entry i is the electrolyser + compressor
// power at request 0.01 * i.  Then hy_reset().  One lane per env replays exactly that: the streams advance by what the
// reference's constructor consumes and the table comes out as the reference's (it depends on the draws whenever a tank
// clamp binds).  Same operations as the step's tail (env_tail), plain f64 divisions.
__global__ void k_compat_ctor_sweep(const DevCtx *__restrict__ ctx, int rng_cur) {
    const HubParams &hp = ctx->hp;
    const EnvArrays &ev = ctx->ev;
    const Tables &tb = ctx->tb;
    const int64_t env = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (env >= hp.n_envs) return;
    CompatStream rs;
    rs.load(ctx->cr, rng_cur, env);
    const int qcap = hp.qcap;
    double *qt = (double *) ev.q_time + (size_t) env * (size_t) qcap, *qm = (double *) ev.q_mass + (size_t) env * (size_t) qcap;
    double *table = (double *) ev.hy_env + (size_t) env * 102u;
    const double cap_mass = hp.cap_mass;
    double cap = hp.init_soc * cap_mass;  // HyStore.__init__ (HYD:99-100)
    int q_len = 0;
    bool stuck = false;
    double fold_t = 0.0, fold_m = 0.0;
    for (int i = 0; i < 101; i++) {
        const int t = i % 96;  // sys_time (HYD:192-193)
        // hvs_step (HYD:253-285)
        const int arrive = (int) tb.cnt_hv[t * kLevels + rs.level()];
        double total_time = stuck ? fold_t : 0.0, total_mass = stuck ? fold_m : 0.0;
        for (int q = 0; q < q_len; q++) {
            total_time += qt[q];
            total_mass += qm[q];
        }
        for (int j = 0; j < arrive; j++) {
            double tn, mn;
            fcev_time_mass(arrive_soc_from(rs.normal_d(7.0, 3.0)), tn, mn);
            if (!stuck && q_len < qcap) {
                qt[q_len] = tn;
                qm[q_len] = mn;
                q_len++;
            }
            total_time += tn;
            total_mass += mn;
        }
        if (total_time > 15.0) {
            int hv_num = 0;
            bool found = false;
            if (!stuck) {
                for (int a = 1; a <= arrive - 1 && !found; a++) {
                    int keep = q_len - a;
                    keep = keep < 0 ? 0 : keep;
                    double part = 0.0;
                    for (int j = 0; j < keep; j++) part += qt[j];
                    if (part <= 15.0) {
                        hv_num = keep;
                        found = true;
                    }
                }
            }
            if (found) {
                for (int j = hv_num; j < q_len; j++) {
                    qt[j - hv_num] = qt[j];
                    qm[j - hv_num] = qm[j];
                }
                q_len -= hv_num;
            } else {
                stuck = true;
                q_len = 0;
                fold_t = total_time;
                fold_m = total_mass;
            }
        } else {
            q_len = 0;
        }
        // hy_step (HYD:160-195) at gen_speed = 0.01 * i
        const double gen_speed = 0.01 * i;
        double must_chg = cap_mass * 0.1 - cap;
        must_chg = must_chg > 0 ? must_chg : 0.0;
        double upper_charge = cap_mass - cap;
        upper_charge = upper_charge > 0 ? upper_charge : 0.0;
        double charge_temp = gen_speed * hp.v_h_max * (15 * 60);
        charge_temp = charge_temp < upper_charge ? charge_temp : upper_charge;
        charge_temp = charge_temp > must_chg ? charge_temp : must_chg;
        double flow = charge_temp / (15 * 60);
        flow = flow < hp.v_h_max ? flow : hp.v_h_max;
        double ele_power = 0.0;
        if (hp.cells != 0.0) {  // Electrolyser.get_power, HYD:38-48
            const double v_H_mass = flow / hp.cells;
            const double v_H_mol = v_H_mass / 2.02;
            const double v_H_L = v_H_mol * hp.v_M;
            const double v_H = v_H_L * 1000 * 60;
            const double temp = v_H * 2 * 96487 / (hp.v_M * 1000 * 60);
            const double power = temp * temp * 0.326 + temp * 1.476;
            ele_power = hp.cells * power / 1000;
        }
        const double cpr_power = ((flow / 2.02) * hp.cpr_w12 / 0.8) / 1000;  // Compressor.generate_W, HYD:74-82
        // sty_step (HYD:104-126)
        cap += flow * 15 * 60;
        double lower_change = cap - 0.1 * cap_mass;
        lower_change = lower_change > 0 ? lower_change : 0.0;
        const double hy_use = total_mass < lower_change ? total_mass : lower_change;
        cap -= hy_use;
        cap -= cap * hp.hydro_loss;
        table[i] = ele_power + cpr_power;
    }
    table[101] = table[100];
    rs.store(ctx->cr, rng_cur, env);
    // hy_reset (HYD:197-208): the step state is re-initialised by chub_reset; the FIFO arrays were scratch here
}

void launch_compat_ctor_sweep(const HubParams &hp, const DevCtx *ctx, int rng_cur, hipStream_t stream) {
    hipLaunchKernelGGL(k_compat_ctor_sweep, dim3((unsigned) ((hp.n_envs + 63) / 64)), dim3(64), 0, stream, ctx, rng_cur);
}

// -------------------------------------------------------------------- random policy (bench / tests)
__global__ void k_random_actions(int64_t n_envs, int64_t env_id0, int act_dim, uint32_t k0, uint32_t k1, uint32_t batch,
                                 float *actions) {
    const int64_t total = n_envs * (int64_t) ((act_dim + 3) / 4);
    const int per = (act_dim + 3) / 4;
    for (int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t) gridDim.x * blockDim.x) {
        const int64_t env = i / per;
        const int j4 = (int) (i % per);
        U4 o = philox4x32_10((uint32_t) j4, 0u, batch, (uint32_t) (env_id0 + env), k0, k1);
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const int j = j4 * 4 + c;
            if (j < act_dim) actions[env * act_dim + j] = (float) (o.v[c] >> 8) * (2.0f / 16777216.0f) - 1.0f;
        }
    }
}

// -------------------------------------------------------------------- introspection: current SoC on demand
// The step kernels keep, per occupied slot, the arrival SoC (COMPAT: the hot record's second word; PHILOX: the class) and the number of car_steps taken since
// (hot.w bits 25-31) instead of storing the SoC every step: the SoC after n steps is the same deterministic chain
// soc -> soc_to_time -> +1 slot -> time_to_soc the step evaluated (CHS.hpp:900-905 / 1065-1070), replayed here with the
// same device functions, so the value is bit for bit the one the step produced.
__global__ void k_replay_soc(const DevCtx *__restrict__ ctx, float *out) {
    const HubParams &hp = ctx->hp;
    const int64_t NS = hp.n_envs * (int64_t) (hp.S[0] + hp.S[1]);
    const int64_t idx = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= NS) return;
    const int St = hp.S[0] + hp.S[1];
    const int k = hp.rng_mode == MODE_PHILOX ? ((int) (idx % St) >= hp.S[0] ? 1 : 0) : (idx >= hp.base[1] ? 1 : 0);
    const bool cp = hp.constant_charging != 0;
    float soc = 0.0f;
    int n = 0;
    bool car;
    if (hp.rng_mode == MODE_PHILOX) {  // arrival SoC of the slot's class, car_steps from the state word
        const uint32_t w0 = ctx->sl.hot[idx];
        car = ps_tl(w0) != 0;
        if (car) {
            soc = ctx->tb.cls_soc0[k][ps_cls(w0)];
            n = (int) ps_n(w0);
        }
    } else {
        const uint32_t w = ctx->sl.hot[4 * idx + 3];
        car = (w & 127u) != 0u;
        if (car) {
            soc = __uint_as_float(ctx->sl.hot[4 * idx + 1]);  // the record's second word: the arrival SoC
            n = (int) (w >> 25);
        }
    }
    for (int i = 0; i < n; i++) {
        float pw;
        if (hp.type[k] == 0) car_step_curves<0>(__fadd_rn(soc_to_time<0>(soc, cp), 1.0f), cp, hp.cc, soc, pw);
        else car_step_curves<1>(__fadd_rn(soc_to_time<1>(soc, cp), 1.0f), cp, hp.cc, soc, pw);
    }
    out[idx] = soc;
}

// PHILOX reset: evs_reset's initial occupancy per (station, env) unit -- init_station_car_number(mu, 3) (CHS.hpp:832-842)
// thinned by the balk test of an empty queue -- drawn once per unit here (every lane of the unit used to redo it),
// handed to k_slot<RESET> through this tick's pk word: arrivals | arrivals that stay << 8.
__global__ void k_reset_levels(const DevCtx *__restrict__ ctx, StepArgs sa) {
    // four lanes per unit: lane q takes the Philox blocks q, q + 4, ... of the unit's arrival words (four arrivals each), the
    // quad adds up.  (One lane per unit walking all its arrivals was a 15-iteration dependent chain at two waves per SIMD:
    // 20 us per reset.)
    const uint32_t tick = sa.tick;
    const HubParams &hp = ctx->hp;
    const Tables &tb = ctx->tb;
    const int64_t N = hp.n_envs;
    const int64_t t = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    const int q = (int) (t & 3);
    int k = 0;
    int64_t env = 0;
    const bool live = range_unit(sa, t >> 2, 2, k, env);  // the units of the envs the call names: both stations of env_lo .. env_hi
    const int64_t u = (int64_t) k * N + env;
    const bool served = live && in_group(sa, env);
    const int S = hp.S[k], mu = S / 2;  // round(charge_number / 2) on ints, CHS.hpp:1276
    PhiloxCtx px{hp.key[0], hp.key[1], CHUB_TICK(hp, tick), (uint32_t) (hp.env_id0 + env)};
    const float cn = __fadd_rn(normal_from_word(tb.normal_icdf, tb.normal_tail, px.block(SITE_INIT, (uint32_t) k, 0).v[0]), (float) mu);
    int n_in = (int) roundf(cn);
    n_in = n_in > mu + 3 ? mu + 3 : (n_in < mu - 3 ? mu - 3 : n_in);
    int true_in = 0;
    // arrival j (word 1 + j of the unit's SITE_ARRIVE stream) stays iff u <= expf(-0.01*(line+j)) and j <= S
    for (int blk = q; served && 4 * blk <= n_in; blk += 4) {
        const U4 b = px.block(SITE_ARRIVE, (uint32_t) k, (uint32_t) blk);
#pragma unroll
        for (int w = 0; w < 4; w++) {
            const int j = 4 * blk + w - 1;
            if (j < 0 || j >= n_in) continue;
            const int thr = (int) tb.thr_balk[j < kBalkTab ? j : kBalkTab - 1];
            true_in += ((int) (b.v[w] % 1000u) <= thr && j <= S) ? 1 : 0;
        }
    }
    true_in += __shfl_xor(true_in, 1);
    true_in += __shfl_xor(true_in, 2);
    if (served && q == 0)
        ctx->st.pk[tick & 1u][u] = ((uint32_t) n_in & 0xFFFFu) | ((uint32_t) true_in << 16);  // n_in: signed 16 bits
}

// fresh launches (StepArgs::fresh): this step's station-level draws, made right in front of the slot kernel -- what the
// previous launch's level blocks would have left in pk[tick & 1]
__global__ void k_draw_levels(const DevCtx *__restrict__ ctx, StepArgs sa) {
    const HubParams &hp = ctx->hp;
    const int64_t N = hp.n_envs;
    int kk = 0;
    int64_t env = 0;
    if (!range_unit(sa, (int64_t) blockIdx.x * blockDim.x + threadIdx.x, 2, kk, env)) return;
    const int64_t u = (int64_t) kk * N + env;
    if (!in_group(sa, env)) return;
    const int t = sa.env_clk ? clk_t(env_clk(sa, N, env)) : sa.t;
    // tape mode: the caller's recorded draws instead of this build's (the 64-bit layout above), decoded the same way
    const int line_now = pkd_line(ctx->st.rec[4u * (uint32_t) u + 3u]);
    ctx->st.pk[sa.tick & 1u][u] = sa.pk_tape ? dk_make(sa.pk_tape[u], line_now, hp.type[kk] == 0)
                                             : draw_decoded_levels(hp, ctx->tb, CHUB_TICK(hp, sa.tick), t, kk, env, line_now, hp.type[kk] == 0);
}

// per-env clocks, a call on a subset of the envs: the clocks of the envs outside the launched range move to the other buffer as they are
__global__ void k_keep_clocks(uint16_t *dst, const uint16_t *src, int64_t n) {
    const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[i];
}

// ------------------------------------------------------------------------------------- launchers
// kernel launch with the dispatch's own start / stop timestamps when asked for (chub_profile_*), a plain launch otherwise
#define CHUB_LAUNCH(kernel, grid, block, stream, ev0, ev1, ...)                                         \
    do {                                                                                                \
        if ((ev0) || (ev1)) hipExtLaunchKernelGGL(kernel, grid, block, 0, stream, ev0, ev1, 0, __VA_ARGS__); \
        else hipLaunchKernelGGL(kernel, grid, block, 0, stream, __VA_ARGS__);                           \
    } while (0)
static inline int64_t blocks_for(int64_t n_envs, int H, int block) {
    const int64_t upb = (int64_t) (block / 64) * (64 / H);
    return (n_envs + upb - 1) / upb;
}
// slot_body_split2: a wave's units lie end to end over its 128 virtual lanes
static inline int64_t blocks_for_split2(int64_t n_envs, int U, int block) {
    const int64_t upb = (int64_t) (block / 64) * (128 / U);
    return (n_envs + upb - 1) / upb;
}

// ev0 / ev1 (may be null): kernel start / stop timestamps of the dispatch itself (hipExtLaunchKernelGGL), what
// chub_profile_* reports -- plain hipEventRecord pairs around a launch also count the gap in front of it
template <bool RESET, int MODE>
static void launch_slot_t(const HubParams &hp, const DevCtx *ctx, const StepArgs &sa, hipStream_t stream, hipEvent_t ev0,
                          hipEvent_t ev1) {
    constexpr int BLOCK = 256;
    // (COMPAT units take exactly S lanes, the PHILOX wave-local kernel's the next power of two: its sums are DPP butterflies)
    const int64_t nb0 = blocks_for(hp.n_envs, MODE == MODE_COMPAT ? hp.U[0] : hp.H[0], BLOCK), nb1 = blocks_for(hp.n_envs, MODE == MODE_COMPAT ? hp.U[1] : hp.H[1], BLOCK);
    const bool big = hp.S[0] > 64 || hp.S[1] > 64;  // a unit of more than 64 piles is a workgroup of its own (k_slot_unit)
    if (MODE == MODE_PHILOX && RESET) hipLaunchKernelGGL(k_reset_levels, dim3((unsigned) ((8 * ((int64_t) sa.env_hi - sa.env_lo + 1) + 255) / 256)), dim3(256), 0, stream, ctx, sa);
    if (MODE == MODE_PHILOX && !big) {
        CHUB_LAUNCH((k_slot<RESET, MODE, BLOCK>), dim3((unsigned) (nb0 + nb1)), dim3(BLOCK), stream, ev0, ev1, ctx, sa, nb0);
    } else if (MODE == MODE_COMPAT && !big && hp.compat_split) {
        // the split step: empties -> the stream walks, one env per lane -> the slots of both stations in one launch
        const bool count_first = !RESET && sa.empt_fresh && !sa.walked;
        if (count_first) {  // of every unit, whatever envs the call names: the counts are then good for whoever is stepped next
            StepArgs all = sa;
            all.env_mask = nullptr;
            CHUB_LAUNCH((k_compat_empties<BLOCK>), dim3((unsigned) (nb0 + nb1)), dim3(BLOCK), stream, ev0, (hipEvent_t) nullptr, ctx, all, nb0);
        }
        const bool walk_now = RESET || !sa.walked;  // (otherwise the walk ran beside the previous step's tails, k_env_walk)
        if (walk_now)
            CHUB_LAUNCH((k_compat_walk<RESET>), dim3((unsigned) ((hp.n_envs + 255) / 256)), dim3(256), stream, count_first ? (hipEvent_t) nullptr : ev0, (hipEvent_t) nullptr, ctx, sa);
        if (CHUB_SPLIT2 && !RESET && !sa.load_mode && hp.U[0] >= 8 && hp.U[1] >= 8) {  // two slots per lane (stations of 8 to 64 piles: at most 8 units per virtual wave)
            const int64_t sb0 = blocks_for_split2(hp.n_envs, hp.U[0], BLOCK), sb1 = blocks_for_split2(hp.n_envs, hp.U[1], BLOCK);
            CHUB_LAUNCH((k_slot_split2<BLOCK>), dim3((unsigned) (sb0 + sb1)), dim3(BLOCK), stream, walk_now ? (hipEvent_t) nullptr : ev0, ev1, ctx, sa, sb0);
        } else
            CHUB_LAUNCH((k_slot_split<RESET, BLOCK>), dim3((unsigned) (nb0 + nb1)), dim3(BLOCK), stream, walk_now ? (hipEvent_t) nullptr : ev0, ev1, ctx, sa, nb0);
    } else {
        for (int k = 0; k < 2; k++) {  // the reference streams are consumed station 0 first, then station 1
            StepArgs s2 = sa;
            s2.station_filter = k;
            hipEvent_t e0 = k == 0 ? ev0 : nullptr, e1 = k == 1 ? ev1 : nullptr;
            if (hp.S[k] > 256) CHUB_LAUNCH((k_slot_unit_any<RESET, MODE>), dim3((unsigned) hp.n_envs), dim3(256), stream, e0, e1, ctx, s2, k);
            else if (hp.S[k] > 64) CHUB_LAUNCH((k_slot_unit<RESET, MODE>), dim3((unsigned) hp.n_envs), dim3(256), stream, e0, e1, ctx, s2, k);
            else CHUB_LAUNCH((k_slot<RESET, MODE, BLOCK>), dim3((unsigned) (k ? nb1 : nb0)), dim3(BLOCK), stream, e0, e1, ctx, s2, nb0);
        }
    }
}

// COMPAT split step, lock-step steps of every env of a handle whose stations take the two-slots-per-lane pass: the slot pass of step i
// (sa) beside the stream walks of step i + 1 (sw: walk_far), k_slot_walk2.  Step i's own walk first where it has not run (the first step
// after a reset or anything else that voided it: near, from the committed streams), the empty-slot counts in front of it where they are
// not the previous pass's.
bool slot_walk2_covers(const HubParams &hp) { return CHUB_SPLIT2 && hp.compat_split && hp.S[0] <= 64 && hp.S[1] <= 64 && hp.U[0] >= 8 && hp.U[1] >= 8; }
void launch_slot_walk2(const HubParams &hp, const DevCtx *ctx, const StepArgs &sa, const StepArgs &sw, hipStream_t stream, hipEvent_t ev0, hipEvent_t ev1) {
    constexpr int BLOCK = 256;
    const int64_t nb0 = blocks_for(hp.n_envs, hp.U[0], BLOCK), nb1 = blocks_for(hp.n_envs, hp.U[1], BLOCK);
    const bool count_first = sa.empt_fresh && !sa.walked;
    if (count_first) {
        StepArgs all = sa;
        all.env_mask = nullptr;
        CHUB_LAUNCH((k_compat_empties<BLOCK>), dim3((unsigned) (nb0 + nb1)), dim3(BLOCK), stream, ev0, (hipEvent_t) nullptr, ctx, all, nb0);
    }
    const bool walk_now = !sa.walked;
    if (walk_now)
        CHUB_LAUNCH((k_compat_walk<false>), dim3((unsigned) ((hp.n_envs + 255) / 256)), dim3(256), stream, count_first ? (hipEvent_t) nullptr : ev0, (hipEvent_t) nullptr, ctx, sa);
    const int64_t sb0 = blocks_for_split2(hp.n_envs, hp.U[0], BLOCK), sb1 = blocks_for_split2(hp.n_envs, hp.U[1], BLOCK);
    if (hp.n_envs <= kWalk2HalfMaxEnvs) {
        const int nwalk = (int) ((hp.n_envs + 31) / 32);
        CHUB_LAUNCH((k_slot_walk2<BLOCK, 32>), dim3((unsigned) (nwalk + sb0 + sb1)), dim3(BLOCK), stream, walk_now ? (hipEvent_t) nullptr : ev0, ev1, ctx, sa, sw, sb0, nwalk);
    } else {
        const int nwalk = (int) ((hp.n_envs + 63) / 64);
        CHUB_LAUNCH((k_slot_walk2<BLOCK, 64>), dim3((unsigned) (nwalk + sb0 + sb1)), dim3(BLOCK), stream, walk_now ? (hipEvent_t) nullptr : ev0, ev1, ctx, sa, sw, sb0, nwalk);
    }
}

static PackedArgs make_packed_args(const HubParams &hp, const StepArgs &sa, const PackedPtrs &pp);

void launch_slot(bool reset, const HubParams &hp, const DevCtx *ctx, const StepArgs &sa, hipStream_t stream,
                 const PackedPtrs &pp, hipEvent_t ev0, hipEvent_t ev1) {
    if (hp.rng_mode == MODE_PHILOX && !reset && (sa.fresh || sa.pk_tape))
        hipLaunchKernelGGL(k_draw_levels, dim3((unsigned) ((2 * ((int64_t) sa.env_hi - sa.env_lo + 1) + 255) / 256)), dim3(256), 0, stream, ctx, sa);
    if (hp.rng_mode == MODE_PHILOX) {
        if (hp.packed && !sa.load_mode) {
            if (reset && !sa.car_tape)  // (tape mode: the caller's occupancy draws are in pk already)
                hipLaunchKernelGGL(k_reset_levels, dim3((unsigned) ((8 * ((int64_t) sa.env_hi - sa.env_lo + 1) + 255) / 256)), dim3(256), 0, stream, ctx, sa);
            const PackedArgs pa = make_packed_args(hp, sa, pp);
            // every workgroup of the batch, or (a call on a subset of the envs) those of the range of envs it names
            const uint32_t nb = sa.env_mask ? (uint32_t) (sa.env_hi / hp.epb - sa.env_lo / hp.epb + 1) : (uint32_t) ((hp.n_envs + hp.epb - 1) / hp.epb);
#define CHUB_PACKED2(TAPE_, RESET_, BIG_, MASKED_, BITS_) \
    do {                                                                                                                              \
        if (hp.pblock == kBigBlock)                                                                                                   \
            CHUB_LAUNCH((k_slot_packed<kBigBlock, kBigSlotsPerLane, TAPE_, RESET_, BIG_, MASKED_, BITS_>), dim3(nb), dim3(kBigBlock), stream, ev0, ev1, ctx, sa, pa); \
        else                                                                                                                          \
            CHUB_LAUNCH((k_slot_packed<kPackedBlock, kSlotsPerLane, TAPE_, RESET_, BIG_, MASKED_, BITS_>), dim3(nb), dim3(kPackedBlock), stream, ev0, ev1, ctx, sa, pa); \
    } while (0)
#define CHUB_PACKED1(TAPE_, RESET_, BIG_, MASKED_) CHUB_PACKED2(TAPE_, RESET_, BIG_, MASKED_, false)
#define CHUB_PACKED(TAPE_, RESET_, BIG_) \
    do {                                                              \
        if (sa.env_mask) CHUB_PACKED1(TAPE_, RESET_, BIG_, true);     \
        else CHUB_PACKED1(TAPE_, RESET_, BIG_, false);                \
    } while (0)
            const bool big = hp.S[0] > 64 || hp.S[1] > 64;
            if (reset && sa.car_tape) {  // tape mode runs in lock-step
                if (big) CHUB_PACKED1(true, true, true, false);
                else CHUB_PACKED1(true, true, false, false);
            } else if (reset) {
                if (big) CHUB_PACKED(false, true, true);
                else CHUB_PACKED(false, true, false);
            } else if (sa.car_tape) {
                if (big) CHUB_PACKED1(true, false, true, false);
                else CHUB_PACKED1(true, false, false, false);
            } else if (sa.act_bits) {  // one bit per pile (lock-step entry points only: no mask)
                if (big) CHUB_PACKED2(false, false, true, false, true);
                else CHUB_PACKED2(false, false, false, false, true);
            } else if (big) {
                CHUB_PACKED(false, false, true);
            } else {
                CHUB_PACKED(false, false, false);
            }
#undef CHUB_PACKED
#undef CHUB_PACKED1
#undef CHUB_PACKED2
        } else if (reset) launch_slot_t<true, MODE_PHILOX>(hp, ctx, sa, stream, ev0, ev1);
        else launch_slot_t<false, MODE_PHILOX>(hp, ctx, sa, stream, ev0, ev1);
        return;
    }
    if (reset) launch_slot_t<true, MODE_COMPAT>(hp, ctx, sa, stream, ev0, ev1);
    else launch_slot_t<false, MODE_COMPAT>(hp, ctx, sa, stream, ev0, ev1);
}

static PackedArgs make_packed_args(const HubParams &hp, const StepArgs &sa, const PackedPtrs &pp) {
    PackedArgs pa;
    for (int k = 0; k < 2; k++) {
        pa.S[k] = (uint32_t) hp.S[k];
        pa.type[k] = (uint32_t) hp.type[k];
    }
    pa.n_envs = (uint32_t) hp.n_envs;
    pa.epb = (uint32_t) hp.epb;
    pa.magic = (1u << 20) / (uint32_t) (hp.S[0] + hp.S[1]) + 1u;
    pa.cls_delta = (uint32_t) ((const char *) pp.cls[1] - (const char *) pp.cls[0]);
    pa.state = (CHUB_G(uint32_t)) pp.hot;
    pa.rec = (CHUB_G(uint32_t)) pp.rec;
    pa.pk = (CHUB_G(const uint32_t)) pp.pk[sa.tick & 1u];
    pa.actions = sa.act_bits ? (CHUB_G(const float)) (const void *) sa.act_bits : (CHUB_G(const float)) sa.actions;
    pa.cls0 = (CHUB_G(const float)) pp.cls[0];
    pa.ttab2 = (CHUB_G(const float)) pp.ttab2;
    pa.car_tape = (CHUB_G(const uint32_t)) sa.car_tape;
    pa.key[0] = hp.key[0];
    pa.key[1] = hp.key[1];
    pa.gid0 = (uint32_t) hp.env_id0;
    pa.tick = sa.tick;
    pa.tick_base = hp.tick_base;
    for (int j = 0; j < 8; j++) pa.late[j] = pp.late8[j];
    pa.env_mask = (CHUB_G(const uint8_t)) sa.env_mask;
    pa.blk0 = sa.env_mask ? (uint32_t) (sa.env_lo / hp.epb) : 0u;
    pa.xcd = hp.xcd ? (uint32_t) ((hp.n_envs + hp.epb - 1) / hp.epb) : 0u;  // (the grid of an unmasked launch: launch_slot)
    pa.tail_act = (CHUB_G(float)) pp.st->tail_act;
    pa.stay8 = (CHUB_G(uint8_t)) pp.stay8;
    return pa;
}

// the whole PHILOX lock-step step as ONE launch (k_step_fused): the caller has checked that the hub shape and the call allow it
void launch_step_fused(const HubParams &hp, const DevCtx *ctx, const StepArgs &sa, hipStream_t stream, const PackedPtrs &pp, hipEvent_t ev0,
                       hipEvent_t ev1) {
    if (sa.fresh || sa.pk_tape) hipLaunchKernelGGL(k_draw_levels, dim3((unsigned) ((2 * ((int64_t) sa.env_hi - sa.env_lo + 1) + 255) / 256)), dim3(256), 0, stream, ctx, sa);
    const PackedArgs pa = make_packed_args(hp, sa, pp);
    TailArgs ta = make_tail_args(*pp.ev, *pp.st, hp, sa, pp, false);
    const uint32_t nb = (uint32_t) ((hp.n_envs + hp.epb - 1) / hp.epb);
    if (sa.tail_tape) {  // tape mode (the caller has checked: station draws, car variates and the tail's variates all come from the tape)
        ta.tail_act = nullptr;
        CHUB_LAUNCH((k_step_fused<kPackedBlock, kSlotsPerLane, false, true>), dim3(nb), dim3(kPackedBlock), stream, ev0, ev1, ctx, sa, pa, ta);
        return;
    }
    if (sa.act_bits) {  // one bit per pile: the tails read their two actions from the caller's [N][2] array
        ta.tail_act = (CHUB_G(const float)) sa.act_tail;
        if (hp.epb <= 64)
            CHUB_LAUNCH((k_step_tailwave<kPackedBlock, kSlotsPerLane, true>), dim3(nb), dim3(kPackedBlock + 64), stream, ev0, ev1, ctx, sa, pa, ta);
        else
            CHUB_LAUNCH((k_step_fused<kPackedBlock, kSlotsPerLane, true>), dim3(nb), dim3(kPackedBlock), stream, ev0, ev1, ctx, sa, pa, ta);
        return;
    }
    ta.tail_act = nullptr;  // the tails read their two actions from the action rows (the workgroup has just had them in cache)
    if (hp.epb <= 64)  // (the tail wave's lanes are the workgroup's envs)
        CHUB_LAUNCH((k_step_tailwave<kPackedBlock, kSlotsPerLane>), dim3(nb), dim3(kPackedBlock + 64), stream, ev0, ev1, ctx, sa, pa, ta);
    else
        CHUB_LAUNCH((k_step_fused<kPackedBlock, kSlotsPerLane>), dim3(nb), dim3(kPackedBlock), stream, ev0, ev1, ctx, sa, pa, ta);
}

// COMPAT lock-step reset / step of a handle whose envs all fit one workgroup (the caller has checked): one launch
// a span of n_steps lock-step steps from `sa` (the span's first step) in one launch: chub_run_steps, PHILOX handles on the one-launch step
void launch_steps_fused(const HubParams &hp, const DevCtx *ctx, const StepArgs &sa, hipStream_t stream, const PackedPtrs &pp, int n_steps, int pc0,
                        int64_t first, const float *const *batches, int n_batches, float *const *packed2, bool piped) {
    if (sa.fresh) hipLaunchKernelGGL(k_draw_levels, dim3((unsigned) ((2 * ((int64_t) sa.env_hi - sa.env_lo + 1) + 255) / 256)), dim3(256), 0, stream, ctx, sa);
    const PackedArgs pa = make_packed_args(hp, sa, pp);
    TailArgs ta = make_tail_args(*pp.ev, *pp.st, hp, sa, pp, false);
    ta.tail_act = nullptr;  // (the tails read their two actions from the action rows, as in k_step_fused)
    SpanArgs sp = {};
    sp.n_steps = n_steps;
    sp.pc0 = pc0;
    sp.first = (uint32_t) (first % (2 * n_batches));  // (only first mod n_batches and first mod 2 matter: 2 * n_batches is a multiple of both)
    sp.n_batches = n_batches;
    sp.D = hp.obs_dim;
    for (int j = 0; j < 8; j++) sp.actions[j] = batches[j < n_batches ? j : 0];
    sp.packed[0] = packed2[0];
    sp.packed[1] = packed2[1];
    for (int q = 0; q < 2; q++) {
        sp.pk[q] = (CHUB_G(const uint32_t)) pp.pk[q];
        sp.drw[q] = (CHUB_G(const uint32_t)) pp.ev->drw[q];
        sp.drw_cnt[q] = (CHUB_G(const uint8_t)) pp.ev->drw_cnt[q];
    }
    sp.price = (CHUB_G(const double)) pp.tb->price;
    sp.pvT = (CHUB_G(const double)) pp.tb->pvT;
    sp.wdT = (CHUB_G(const double)) pp.tb->wdT;
    sp.sin96 = (CHUB_G(const double)) pp.tb->sin96;
    const uint32_t nb = (uint32_t) ((hp.n_envs + hp.epb - 1) / hp.epb);
    if (piped)  // (the caller has checked: at most 64 envs per workgroup -- the tail wave's lanes are the workgroup's envs)
        hipLaunchKernelGGL((k_steps_piped<kPackedBlock, kSlotsPerLane>), dim3(nb), dim3(kPackedBlock + 64), 0, stream, ctx, sa, pa, ta, sp);
    else
        hipLaunchKernelGGL((k_steps_fused<kPackedBlock, kSlotsPerLane>), dim3(nb), dim3(kPackedBlock), 0, stream, ctx, sa, pa, ta, sp);
}

void launch_compat_small(bool reset, const HubParams &hp, const DevCtx *ctx, const StepArgs &sa, hipStream_t stream, const PackedPtrs &pp) {
    const TailArgs ta = make_tail_args(*pp.ev, *pp.st, hp, sa, pp, reset);
    if (reset) hipLaunchKernelGGL(k_compat_small<true>, dim3(1), dim3(kCompatSmallBlock), 0, stream, ctx, sa, ta);
    else hipLaunchKernelGGL(k_compat_small<false>, dim3(1), dim3(kCompatSmallBlock), 0, stream, ctx, sa, ta);
}

void launch_env(bool reset, const HubParams &hp, const DevCtx *ctx, const StepArgs &sa, hipStream_t stream, hipEvent_t ev0,
                hipEvent_t ev1, const PackedPtrs &pp) {
    const TailArgs ta = make_tail_args(*pp.ev, *pp.st, hp, sa, pp, reset);
    // the tail blocks of every env, or (per-env clocks) of the range of envs the call names
    const int nb_env = sa.env_clk ? sa.env_hi / kEnvBlock - sa.env_lo / kEnvBlock + 1 : (int) ((hp.n_envs + kEnvBlock - 1) / kEnvBlock);
    if (sa.env_clk && sa.env_mask) {
        // the clocks are double-buffered by launch parity and the tail moves those of the envs of its blocks to the other buffer; the
        // envs outside the range keep theirs: copied here
        const int64_t N = hp.n_envs;
        hipLaunchKernelGGL(k_keep_clocks, dim3((unsigned) ((N + 255) / 256)), dim3(256), 0, stream, sa.env_clk + (int64_t) ((sa.tick + 1u) & 1u) * N,
                           (const uint16_t *) sa.env_clk + (int64_t) (sa.tick & 1u) * N, N);
    }
    if (hp.rng_mode == MODE_PHILOX) {
        // + the level-draw workgroups: next step's state-independent variates (3 lanes per env: two stations, one env)
        const unsigned nb = (unsigned) nb_env + (unsigned) ((3 * ((int64_t) sa.env_hi - sa.env_lo + 1) + kEnvBlock - 1) / kEnvBlock);
        if (sa.tail_tape) {  // tape mode (lock-step): the tail's variates from the caller
            if (reset) CHUB_LAUNCH((k_env<true, MODE_PHILOX, false, true>), dim3(nb), dim3(kEnvBlock), stream, ev0, ev1, ctx, sa, ta, nb_env);
            else CHUB_LAUNCH((k_env<false, MODE_PHILOX, false, true>), dim3(nb), dim3(kEnvBlock), stream, ev0, ev1, ctx, sa, ta, nb_env);
        } else if (sa.env_clk) {  // per-env clocks: its own instantiation
            if (reset) CHUB_LAUNCH((k_env<true, MODE_PHILOX, true>), dim3(nb), dim3(kEnvBlock), stream, ev0, ev1, ctx, sa, ta, nb_env);
            else CHUB_LAUNCH((k_env<false, MODE_PHILOX, true>), dim3(nb), dim3(kEnvBlock), stream, ev0, ev1, ctx, sa, ta, nb_env);
        } else if (reset) CHUB_LAUNCH((k_env<true, MODE_PHILOX, false>), dim3(nb), dim3(kEnvBlock), stream, ev0, ev1, ctx, sa, ta, nb_env);
        else CHUB_LAUNCH((k_env<false, MODE_PHILOX, false>), dim3(nb), dim3(kEnvBlock), stream, ev0, ev1, ctx, sa, ta, nb_env);
    } else {
        if (sa.env_clk) {  // per-env clocks
            if (reset) CHUB_LAUNCH((k_env<true, MODE_COMPAT, true>), dim3((unsigned) nb_env), dim3(kEnvBlock), stream, ev0, ev1, ctx, sa, ta, nb_env);
            else CHUB_LAUNCH((k_env<false, MODE_COMPAT, true>), dim3((unsigned) nb_env), dim3(kEnvBlock), stream, ev0, ev1, ctx, sa, ta, nb_env);
        } else if (reset) CHUB_LAUNCH((k_env<true, MODE_COMPAT, false>), dim3((unsigned) nb_env), dim3(kEnvBlock), stream, ev0, ev1, ctx, sa, ta, nb_env);
        else CHUB_LAUNCH((k_env<false, MODE_COMPAT, false>), dim3((unsigned) nb_env), dim3(kEnvBlock), stream, ev0, ev1, ctx, sa, ta, nb_env);
    }
}

// COMPAT split step, lock-step: the tails of step `sa` + the walks of the step after it (`sw`: its clock and tick)
void launch_env_walk(const HubParams &hp, const DevCtx *ctx, const StepArgs &sa, const StepArgs &sw, hipStream_t stream, hipEvent_t ev0, hipEvent_t ev1,
                     const PackedPtrs &pp) {
    const TailArgs ta = make_tail_args(*pp.ev, *pp.st, hp, sa, pp, false);
    const int nb_env = (int) ((hp.n_envs + kEnvBlock - 1) / kEnvBlock);
    const unsigned nb = (unsigned) nb_env + (unsigned) ((hp.n_envs + 255) / 256);
    CHUB_LAUNCH(k_env_walk, dim3(nb), dim3(kEnvBlock), stream, ev0, ev1, ctx, sa, ta, sw, nb_env);
}

__global__ void k_tick_advance(uint32_t *tick_base, uint32_t by) { *tick_base += by; }


// chub_step_bits: the pile decisions arrive as one bit per pile (what action_to_real makes of the action row, MGR:384-393) and
// the two tail actions as floats; this writes the [N, S + 2] f32 action rows the step kernels read: +1 / -1 for the piles (any
// value on the same side of the threshold gives the same step), the tail as it is
__global__ void k_expand_bits(int64_t n_envs, int S, int W, const uint64_t *__restrict__ bits, const float *__restrict__ tail,
                              float *__restrict__ actions) {
    const uint32_t A = (uint32_t) S + 2u;
    const uint32_t total = (uint32_t) n_envs * A;  // n_envs * (piles + 2) < 2^31 (checked at create)
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const uint32_t env = i / A;
        const uint32_t j = i - env * A;
        float v;
        if (j < (uint32_t) S) v = ((bits[(size_t) env * (size_t) W + (j >> 6)] >> (j & 63u)) & 1ull) ? 1.0f : -1.0f;
        else v = tail[env * 2u + (j - (uint32_t) S)];
        actions[i] = v;
    }
}
void launch_expand_bits(const HubParams &hp, const uint64_t *d_bits, const float *d_tail, float *d_actions, hipStream_t stream) {
    const int S = hp.S[0] + hp.S[1];
    const int64_t total = hp.n_envs * (int64_t) (S + 2);
    int64_t nb = (total + 255) / 256;
    if (nb > 16384) nb = 16384;
    hipLaunchKernelGGL(k_expand_bits, dim3((unsigned) nb), dim3(256), 0, stream, hp.n_envs, S, (S + 63) / 64, d_bits, d_tail, d_actions);
}

// entering per-env clocks: every env starts from the handle's lock-step clock
__global__ void k_fill_clocks(uint16_t *dst, int64_t n, uint16_t value) {
    const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = value;
}
void launch_fill_clocks(uint16_t *dst, int64_t n, uint16_t value, hipStream_t stream) {
    hipLaunchKernelGGL(k_fill_clocks, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, stream, dst, n, value);
}
void launch_keep_clocks(uint16_t *dst, const uint16_t *src, int64_t n, hipStream_t stream) {
    hipLaunchKernelGGL(k_keep_clocks, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, stream, dst, src, n);
}
void launch_tick_advance(uint32_t *tick_base, uint32_t by, hipStream_t stream) {
    hipLaunchKernelGGL(k_tick_advance, dim3(1), dim3(1), 0, stream, tick_base, by);
}

// chub_create, COMPAT handles: every entry of Tables::ttab against the device's own soc_to_time(uniform_level(l, 80, 100)) -- the expression
// add_car evaluates (make_car's callers) -- bit for bit; *mismatch counts the entries that differ
__global__ void k_check_ttab(const DevCtx *__restrict__ ctx, uint32_t *mismatch) {
    const int i = (int) (blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= 2 * kLevels) return;
    const int k = i / kLevels, lev = i - k * kLevels;
    const bool cp = ctx->hp.constant_charging != 0;
    const float target = uniform_level(lev, 80.0f, 100.0f);
    const float t = ctx->hp.type[k] == 0 ? soc_to_time<0>(target, cp) : soc_to_time<1>(target, cp);
    if (__float_as_uint(t) != __float_as_uint(ctx->tb.ttab[k][lev])) atomicAdd(mismatch, 1u);
}
void launch_check_ttab(const DevCtx *ctx, uint32_t *d_mismatch, hipStream_t stream) {
    hipLaunchKernelGGL(k_check_ttab, dim3((2 * kLevels + 255) / 256), dim3(256), 0, stream, ctx, d_mismatch);
}

void launch_replay_soc(const HubParams &hp, const DevCtx *ctx, float *d_out, hipStream_t stream) {
    const int64_t NS = hp.n_envs * (int64_t) (hp.S[0] + hp.S[1]);
    hipLaunchKernelGGL(k_replay_soc, dim3((unsigned) ((NS + 255) / 256)), dim3(256), 0, stream, ctx, d_out);
}

void launch_random_actions(const HubParams &hp, uint64_t key, uint32_t batch, float *d_actions, hipStream_t stream) {
    const int per = (hp.act_dim + 3) / 4;
    int64_t total = hp.n_envs * per;
    int64_t nb = (total + 255) / 256;
    if (nb > 8192) nb = 8192;
    if (nb < 1) nb = 1;
    hipLaunchKernelGGL(k_random_actions, dim3((unsigned) nb), dim3(256), 0, stream, hp.n_envs, hp.env_id0, hp.act_dim,
                       (uint32_t) key, (uint32_t) (key >> 32), batch, d_actions);
}

}  // namespace chub
