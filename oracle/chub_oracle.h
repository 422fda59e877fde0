/* TEST INFRASTRUCTURE -- NOT PRODUCT CODE.
 *
 * chub_oracle: a scalar CPU restatement, in plain C, of the reference's per-step path
 * (XJTU-Power-AI/charginghub-env @ 2025-05-23).  Every function cites the reference file:line it
 * follows.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library, and only as the checker / reported baseline -- the product (charginghub-env_amd/) never
 * links, imports or executes anything under oracle/.
 *
 * Parity status: PINNED.  (1) The C++ half (rows a1-a17 of SURVEY.md section 8) is checked against
 * the real reference compiled from its own sources (oracle/_ref/libchs_ref.so, recipe in
 * oracle/Makefile) by tests/test_oracle_vs_ref.py in the build container, and against golden
 * vectors generated from it (tests/golden/, generator oracle/gen/).  (2) The Python half (a18-a28)
 * is checked against golden trajectories produced by importing the unmodified reference .py files
 * in the build container (oracle/gen/gen_env_golden.py).
 *
 * Abbreviations: CHS = evcssp_env_cpp/envs/lion_cpp20/SCP_Base/CHS.hpp, MGR = envs/evcssp_manager.py,
 * AGG = lion_cpp20/Aggregator_Simple.py, HYD = lion_cpp20/hydro_sys.py, REN = lion_cpp20/renewable.py.
 */
#ifndef CHUB_ORACLE_H
#define CHUB_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#ifndef ORC_MAX_PILES
#define ORC_MAX_PILES 256 /* piles per station the plain arrays below hold; liboracle_big.so is the same source with -DORC_MAX_PILES=4096 (stations of more than 256 piles: tests/test_gpu_big_stations.py, the env_big_300_270 fixture) */
#endif
#define ORC_CDF_ROWS 96
#define ORC_CDF_COLS 301
#define ORC_SOC_LEVELS 2048 /* PHILOX mode: equiprobable levels of the EV arrival SoC */
#define ORC_QCAP 1024 /* FCEV waiting list: the reference list is unbounded (HYD:264-265); this plain array is long enough for every test (saturating, flagged in q_overflow: tests assert the flag stays 0) */

enum { ORC_FAST = 0, ORC_SLOW = 1 };
enum { ORC_RNG_COMPAT = 0, ORC_RNG_PHILOX = 1 };

/* draw-site tags (Philox counter word 1 = tag << 16 | index); compat mode ignores them */
enum {
    ORC_PU_ARRIVE = 1, ORC_PU_INIT = 2, ORC_PU_RENEGE = 3, ORC_PU_BALK = 4, ORC_PU_SOC = 5,
    ORC_PU_TGT = 6, ORC_PU_LATE = 7, ORC_PU_HV = 8, ORC_PU_HVSOC = 9, ORC_PU_OU = 10, ORC_PU_DAY = 11
};

typedef struct {
    int mode;
    /* compat: glibc TYPE_3 additive-feedback ring (CHS:35-44 -> rand()) + minstd_rand0 (CHS:25) */
    uint32_t g[31];
    int gf, gr;
    uint32_t minstd;
    /* philox4x32-10: key = seed, counter = (block, tag<<16|index, tick, env) */
    uint32_t key[2];
    uint32_t env_id;
    uint32_t tick;
} orc_rng;

typedef struct {
    int type, n, wait, constant_charging;
    /* Station::situation, CHS:204-231 */
    float car[ORC_MAX_PILES], charge[ORC_MAX_PILES], emergency[ORC_MAX_PILES], assign[ORC_MAX_PILES];
    float power[ORC_MAX_PILES], soc[ORC_MAX_PILES], init_soc[ORC_MAX_PILES], target_soc[ORC_MAX_PILES];
    /* ChargePositionBase, CHS:237-246 */
    float p_arrive_soc[ORC_MAX_PILES], p_target_soc[ORC_MAX_PILES], p_current_soc[ORC_MAX_PILES];
    float p_current_power[ORC_MAX_PILES];
    int stay_time[ORC_MAX_PILES], already[ORC_MAX_PILES];
    int time_hole, line, flow_in_last, has_flow;
    int empty_list[ORC_MAX_PILES], empty_number;
    float load_assigned;
    float min_power, max_power, charge_power;
    int car_number;
    float constant_power, transformer_limit;
    int slot_base; /* hub-global index of slot 0 (Philox tags) */
    int index;     /* 0/1 within the hub */
    int exact_sums; /* 0: reference's sequential f32 sums (CHS:1244-1255); 1: order-independent integer sums in 2^-19 kW (PHILOX mode) */
} orc_station;

typedef struct {
    int piles[2];
    int type[2];
    int constant_charging;
    double hydro_prod_rate;  /* m^3/h, default 430 (HYD:140-143) */
    double hydro_store_vlt;  /* m^3, default 5000 (HYD:96) */
    double init_soc;
    double fc_max_power;     /* HFC cell_number, default 100 (HYD:401-404) */
    double fcev_permeate;
    double renew_fluctuate, price_fluctuate, hydro_loss;
} orc_config;

typedef struct {
    float cdf[ORC_CDF_ROWS][ORC_CDF_COLS]; /* parsed by orc_parse_float (CHS:138-155), widened on use */
    int cdf_rows, cdf_cols;
    double price[96];
    double pv[100][96];
    double wd[150][96];
    /* PHILOX-mode sampling tables (tools/gen_tables.py): inverse CDF of clip(N(7,3),1,10), thresholds of mk_late_time */
    float soc_d_icdf[4097];
    uint32_t late_thr[16];
    float normal_icdf[4097], normal_tail[4097]; /* two-level inverse CDF of N(0,1) */
} orc_tables;

typedef struct {
    orc_config cfg;
    const orc_tables *tab;
    orc_rng rng;
    orc_station st[2];
    /* HySystem / HyStore / Electrolyser / HyFCEVStation / HFC (HYD) */
    double v_h_max, cap_mass, capacity, store_soc;
    long cell_number;
    double cpr_kw_per_gs;
    double q_time[ORC_QCAP], q_mass[ORC_QCAP];
    int q_len, q_overflow;
    int hv_line, hv_num, hv_arrive;
    double total_mass_need, hy_flow_speed, ele_power, cpr_power, all_power_second;
    double hy_use, not_meet, hy_to_use, fc_power, hy_act;
    double hy_table[102], hy_table_in[102];
    int sys_time;
    /* ReNew / OU (REN) + price (MGR:344-361) */
    int pv_day, wd_day;
    double ou_pv, ou_wd, ou_price;
    double price_noise_part; /* self.price_next, MGR:356 */
    int price_count;
    double price_last;       /* env_aggregator.price[-1], AGG:147,171 */
    int agg_time;
    double re_pv, re_wd;
    double price_mean, price_std;
    /* real_state (MGR:364-372) */
    double time_now, price_next;
    /* last-step telemetry (MGR:183-269) */
    double used_renew, ev_list[2], hydrogen_power_grid, income, reward;
    double ev_net[2], ev_sum_net, price_now; /* ev_power_list / ev_power_sum after the fuel cell, real_state[1] as the step found it */
    int obs_dim;
} orc_env;

/* ---- data ---- */
float orc_parse_float(const char *s, int len);
int orc_load_cdf_csv(const char *path, orc_tables *t);
int orc_load_f64(const char *path, double *dst, long count);
int orc_arrival_index(const orc_tables *t, int time, int k);           /* CHS:731-743 with u = k/999 */
int orc_count_fast(int n);                                             /* CHS:751-756 */
int orc_count_slow(int n);                                             /* CHS:758-763 */
int orc_count_hv(int n, float possible_in, float permeability);        /* CHS:765-780 */
float orc_uniform_level(int k, float a, float b);                      /* CHS:35-44 given rand()%1000 == k */

/* ---- RNG ---- */
void orc_rng_seed_compat(orc_rng *r, uint32_t glibc_seed, uint32_t minstd_seed);
void orc_rng_seed_philox(orc_rng *r, uint64_t seed, uint32_t env_id);
uint32_t orc_glibc_rand(orc_rng *r);
uint32_t orc_minstd_next(orc_rng *r);
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
int orc_draw_k(orc_rng *r, int tag, int index, int j);
float orc_mk_soc(orc_rng *r);                                          /* CHS:804-814, reference streams */
int orc_mk_late_time(orc_rng *r);                                      /* CHS:816-830 ("slow" law), reference streams */
float orc_soc_from_word(const orc_tables *t, uint32_t w);              /* PHILOX mode: mk_soc from one 32-bit uniform (FCEV) */
float orc_soc_level_value(const orc_tables *t, uint32_t level);       /* PHILOX mode: EV arrival SoC of level 0..ORC_SOC_LEVELS-1 */
float orc_soc_level_from_word(const orc_tables *t, uint32_t w);       /* PHILOX mode: EV arrival SoC from one uniform */
int orc_late_from_word(const orc_tables *t, uint32_t w);               /* PHILOX mode: mk_late_time from one uniform */
int orc_init_station_car_number(orc_rng *r, const orc_tables *t, int station, int mu); /* CHS:832-842 */
float orc_normal_from_word(const orc_tables *t, uint32_t w);           /* PHILOX mode: N(0,1) from one 32-bit uniform */
void orc_rng_export_glibc128(const orc_rng *r, unsigned char *buf132); /* glibc initstate layout + minstd */
void orc_rng_import_glibc128(orc_rng *r, const unsigned char *buf132);

/* ---- charge curves (CHS:467-590, 593-726); which: 0 time_to_power 1 time_to_soc 2 soc_to_time ---- */
float orc_curve_slow(int which, float x, int constant_power);
float orc_curve_fast(int which, float x, int constant_power);

/* ---- station (CHS:1106-1726) ---- */
void orc_station_init(orc_station *s, int type, int piles, int wait, int constant_charging, int index, int slot_base);
void orc_station_reset(orc_station *s, orc_rng *r, const orc_tables *t);
void orc_station_step(orc_station *s, orc_rng *r, const orc_tables *t, const float *actions);
void orc_station_step_load(orc_station *s, orc_rng *r, const orc_tables *t, float load);

/* ---- H2 pieces (HYD) ---- */
double orc_j2601_target_pressure(double p0);                           /* HYD:338-388 */
void orc_j2601_time_mass(double p0, double *time_need, double *mass_need); /* HYD:308-321 */
double orc_electrolyser_power(double flow_gs, long cells);             /* HYD:38-48 */
long orc_electrolyser_cells(double v_h_max);                           /* HYD:22-24 */
double orc_compressor_kw(double flow_gs);                              /* HYD:74-82 */

/* ---- full env (MGR + AGG + HYD + REN) ---- */
void orc_env_init(orc_env *e, const orc_config *cfg, const orc_tables *t);
/* exo_days: {pv_day, wd_day} or NULL (Philox mode draws them); exo_z: {z_pv, z_wd, z_price} or NULL */
void orc_env_reset(orc_env *e, const int *exo_days, const double *exo_z, double *obs);
void orc_env_step(orc_env *e, const float *action, const double *exo_z, double *obs, double *reward, int *done);
void orc_env_step_load(orc_env *e, const float *action, const double *exo_z, double *obs, double *reward, int *done);
int orc_env_obs_dim(const orc_config *cfg);

/* ---- vector front-end used by tests and by bench.py's cpu_baseline ---- */
typedef struct orc_vec orc_vec;
orc_vec *orc_vec_create(const orc_config *cfg, const orc_tables *t, long n_envs, long env_id0, int rng_mode,
                        uint64_t seed);
void orc_vec_destroy(orc_vec *v);
orc_env *orc_vec_env(orc_vec *v, long i);
void orc_vec_reset(orc_vec *v, const int *exo_days, const double *exo_z, double *obs);
void orc_vec_step(orc_vec *v, const float *actions, const double *exo_z, double *obs, double *reward,
                  unsigned char *done, int n_threads);
void orc_vec_step_load(orc_vec *v, const float *actions, const double *exo_z, double *obs, double *reward,
                       unsigned char *done, int n_threads);
long orc_sizeof_env(void);
long orc_check_canon_division(long n, unsigned long long seed); /* see chub_oracle.c */

/* ---- accessors for the ctypes tests ---- */
orc_tables *orc_tables_load(const char *data_dir);
void orc_tables_free(orc_tables *t);
float orc_tables_cdf(const orc_tables *t, int r, int c);
orc_rng *orc_rng_alloc(void);
void orc_rng_free(orc_rng *r);
void orc_rng_set_tick(orc_rng *r, uint32_t tick);
orc_station *orc_station_alloc(void);
void orc_station_free(orc_station *s);
void orc_station_scalars(const orc_station *s, double *out8);
void orc_station_slots(const orc_station *s, float *out /* [9][n] */);
orc_env *orc_env_alloc(void);
void orc_env_free(orc_env *e);
orc_station *orc_env_station(orc_env *e, int k);
orc_rng *orc_env_rng(orc_env *e);
void orc_env_hy_table(const orc_env *e, double *out102);
int orc_env_telemetry(const orc_env *e, double *out24);
int orc_env_q_overflow(const orc_env *e); /* 1 once the FCEV waiting list outgrew ORC_QCAP */
void orc_env_set_hy_table(orc_env *e, const double *in102);
void orc_env_init_compat_ctor(orc_env *e, const orc_config *cfg, const orc_tables *t, uint32_t glibc_seed,
                              uint32_t minstd_seed);

#ifdef __cplusplus
}
#endif
#endif
