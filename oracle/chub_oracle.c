/* TEST INFRASTRUCTURE -- NOT PRODUCT CODE.  See chub_oracle.h for scope, citations key and parity status.
 *
 * Scalar restatement of the reference's hot path.  It deliberately keeps the reference's shape
 * (duplicate calculate_output calls, float "situation" flags, serial RNG consumption order) so that
 * it can be diffed against the reference line by line; speed is irrelevant here.
 *
 * Mixed-precision rule used everywhere: a `float` variable / parameter /
 * return value is one rounding to f32; float (op) double-literal is f64; int*float is f32;
 * pow/exp/log are f64; logf/expf are f32.  Compile with -ffp-contract=off.
 */
#define _GNU_SOURCE
#include "chub_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ======================================================================================== data */

/* CHS:138-155 Read2Vector::string_to_float -- float accumulator, d *= 0.1 in double then narrowed */
float orc_parse_float(const char *s, int len) {
    int i = 0;
    float sum = 0;
    while (i < len) {
        if (s[i] == '.') break;
        sum = sum * 10 + s[i] - '0'; /* ((sum*10f) + (float)ch) - 48f, all f32 */
        ++i;
    }
    ++i;
    float t, d = 1;
    while (i < len) {
        d *= 0.1; /* f32 <- f64 product */
        t = s[i] - '0';
        sum += t * d;
        ++i;
    }
    return sum;
}

/* CHS:96-136 Read2Vector::read + file_to_string: only digits and '.' are kept, ',' separates */
int orc_load_cdf_csv(const char *path, orc_tables *t) {
    FILE *f = fopen(path, "rb");
    if (!f) return 1;
    char *line = NULL;
    size_t cap = 0;
    ssize_t n;
    int row = 0;
    while ((n = getline(&line, &cap, f)) > 0 && row < ORC_CDF_ROWS) {
        char cur[128];
        int cl = 0, col = 0;
        for (ssize_t p = 0; p < n; p++) {
            char c = line[p];
            if ((c >= '0' && c <= '9') || c == '.') {
                if (cl < 127) cur[cl++] = c;
            } else if (c == ',' && cl > 0) {
                if (col < ORC_CDF_COLS) t->cdf[row][col++] = orc_parse_float(cur, cl);
                cl = 0;
            }
        }
        if (cl > 0 && col < ORC_CDF_COLS) t->cdf[row][col++] = orc_parse_float(cur, cl);
        if (row == 0) t->cdf_cols = col;
        else if (col != t->cdf_cols) { free(line); fclose(f); return 2; }
        row++;
    }
    free(line);
    fclose(f);
    t->cdf_rows = row;
    return (row == ORC_CDF_ROWS && t->cdf_cols == ORC_CDF_COLS) ? 0 : 3;
}

int orc_load_f64(const char *path, double *dst, long count) {
    FILE *f = fopen(path, "rb");
    if (!f) return 1;
    long got = (long) fread(dst, sizeof(double), (size_t) count, f);
    fclose(f);
    return got == count ? 0 : 2;
}

/* CHS:35-44 RandomUtil::uniform_rand given rand() % 1000 == k */
float orc_uniform_level(int k, float a, float b) {
    int N = 999;
    float tr = k / (float) (N);
    tr = tr * (b - a) + a;
    return tr;
}

/* CHS:731-743 give_car_number_wrt_poisson: first index whose CDF value (double) >= u (float) */
int orc_arrival_index(const orc_tables *t, int time, int k) {
    int car_max = 300;
    float compare_possible = orc_uniform_level(k, 0, 1);
    for (int j = 0; j < t->cdf_cols; j++) {
        double item = (double) t->cdf[time][j];
        if (item >= compare_possible) return j;
    }
    return car_max;
}

/* CHS:751-756 */
int orc_count_fast(int n) {
    float possible_in = 0.15;
    float permeability = 0.2;
    float po_ev_number = possible_in * permeability * n;
    return (int) roundf(po_ev_number);
}
/* CHS:758-763 */
int orc_count_slow(int n) {
    float possible_in = 0.1;
    float permeability = 0.2;
    float po_ev_number = possible_in * permeability * n;
    return (int) roundf(po_ev_number);
}
/* CHS:765-780 */
int orc_count_hv(int n, float possible_in_, float permeability_) {
    float hv_possible_in, hv_permeability;
    if (possible_in_ > 1) hv_possible_in = 0.3; else hv_possible_in = possible_in_;
    if (permeability_ > 1) hv_permeability = 0.01; else hv_permeability = permeability_;
    return (int) roundf(hv_possible_in * hv_permeability * n);
}

/* ========================================================================================= RNG */

/* glibc random_r.c srandom_r for TYPE_3 (deg 31, sep 3): what rand() is when srand() was never
 * called (seed 1), CHS:35-44 with seed_rand=False (CHS:27, MGR:28). */
void orc_rng_seed_compat(orc_rng *r, uint32_t glibc_seed, uint32_t minstd_seed) {
    memset(r, 0, sizeof *r);
    r->mode = ORC_RNG_COMPAT;
    if (glibc_seed == 0) glibc_seed = 1;
    int32_t word = (int32_t) glibc_seed;
    r->g[0] = (uint32_t) word;
    for (int i = 1; i < 31; i++) {
        long hi = word / 127773;
        long lo = word % 127773;
        word = (int32_t) (16807 * lo - 2836 * hi);
        if (word < 0) word += 2147483647;
        r->g[i] = (uint32_t) word;
    }
    r->gf = 3;
    r->gr = 0;
    for (int i = 0; i < 310; i++) (void) orc_glibc_rand(r);
    /* std::minstd_rand0::seed: x = s mod m, 0 -> 1 */
    uint32_t x = minstd_seed % 2147483647u;
    if (x == 0) x = 1;
    r->minstd = x;
}

uint32_t orc_glibc_rand(orc_rng *r) {
    r->g[r->gf] += r->g[r->gr];
    uint32_t result = r->g[r->gf] >> 1;
    if (++r->gf >= 31) r->gf = 0;
    if (++r->gr >= 31) r->gr = 0;
    return result;
}

/* std::minstd_rand0 (libstdc++ default_random_engine, CHS:25): x <- 16807 x mod (2^31 - 1) */
uint32_t orc_minstd_next(orc_rng *r) {
    r->minstd = (uint32_t) (((uint64_t) r->minstd * 16807u) % 2147483647u);
    return r->minstd;
}

/* glibc initstate()/setstate() buffer layout: int header = (rptr - state) * 5 + 3, then 31 words;
 * followed here by the minstd word.  Lets tests move a stream between oracle and oracle/_ref. */
void orc_rng_export_glibc128(const orc_rng *r, unsigned char *buf) {
    int32_t hdr = r->gr * 5 + 3;
    memcpy(buf, &hdr, 4);
    memcpy(buf + 4, r->g, 31 * 4);
    memcpy(buf + 128, &r->minstd, 4);
}
void orc_rng_import_glibc128(orc_rng *r, const unsigned char *buf) {
    int32_t hdr;
    memcpy(&hdr, buf, 4);
    r->mode = ORC_RNG_COMPAT;
    r->gr = hdr / 5;
    r->gf = (r->gr + 3) % 31;
    memcpy(r->g, buf + 4, 31 * 4);
    memcpy(&r->minstd, buf + 128, 4);
}

/* Philox4x32-10 (Salmon et al., SC'11), the production-mode generator named by the north star */
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int i = 0; i < 10; i++) {
        uint64_t p0 = (uint64_t) 0xD2511F53u * c0;
        uint64_t p1 = (uint64_t) 0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t) (p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t) p1;
        uint32_t n2 = (uint32_t) (p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t) p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

void orc_rng_seed_philox(orc_rng *r, uint64_t seed, uint32_t env_id) {
    memset(r, 0, sizeof *r);
    r->mode = ORC_RNG_PHILOX;
    r->key[0] = (uint32_t) seed;
    r->key[1] = (uint32_t) (seed >> 32);
    r->env_id = env_id;
    r->tick = 0;
}

static void philox_block(const orc_rng *r, int tag, int index, uint32_t block, uint32_t out[4]) {
    uint32_t ctr[4] = {block, ((uint32_t) tag << 16) | (uint32_t) index, r->tick, r->env_id};
    orc_philox4x32_10(ctr, r->key, out);
}

/* one 1000-level uniform: compat = rand() % 1000 (CHS:41); philox = word j of the site's stream */
int orc_draw_k(orc_rng *r, int tag, int index, int j) {
    if (r->mode == ORC_RNG_COMPAT) return (int) (orc_glibc_rand(r) % 1000u);
    uint32_t o[4];
    philox_block(r, tag, index, (uint32_t) (j >> 2), o);
    return (int) (o[j & 3] % 1000u);
}

/* libstdc++ generate_canonical<double,53>(minstd_rand0): two draws (random.tcc:3348-3380) */
static double canon_double_minstd(orc_rng *r) {
    const double R = 2147483646.0;
    double sum = 0, tmp = 1;
    sum += (double) (orc_minstd_next(r) - 1u) * tmp;
    tmp *= R;
    sum += (double) (orc_minstd_next(r) - 1u) * tmp;
    tmp = (double) ((long double) tmp * (long double) R);
    double ret = sum / tmp;
    if (ret >= 1.0) ret = nextafter(1.0, 0.0);
    return ret;
}
/* generate_canonical<float,24>: one draw */
static float canon_float_minstd(orc_rng *r) {
    float sum = (float) (orc_minstd_next(r) - 1u);
    float tmp = (float) 2147483646.0L;
    float ret = sum / tmp;
    if (ret >= 1.0f) ret = nextafterf(1.0f, 0.0f);
    return ret;
}

/* std::normal_distribution<double>, freshly constructed per call (CHS:805): Marsaglia polar,
 * returns y*mult, discards the x twin (random.tcc:1802-1835) */
static double normal_double_compat(orc_rng *r, double mean, double sd) {
    double x, y, r2;
    do {
        x = 2.0 * canon_double_minstd(r) - 1.0;
        y = 2.0 * canon_double_minstd(r) - 1.0;
        r2 = x * x + y * y;
    } while (r2 > 1.0 || r2 == 0.0);
    double mult = sqrt(-2 * log(r2) / r2);
    double ret = y * mult;
    return ret * sd + mean;
}
/* tests/test_host_cpu.py: the device's stream walk divides generate_canonical<double>'s sum by the constant R * R as q = sum * RN(1 / (R * R)),
 * one exact residual, one correction (two fmas: Markstein 1990 -- with the correctly rounded reciprocal and q within an ulp that IS the
 * correctly rounded quotient).  Here the identity is checked against the division itself on n dividends of the shape the walk produces
 * (a + b * R, a and b draws of minstd_rand0 minus one; the first n / 16 with b near 0, the next n / 16 with b near its maximum).
 * -> the number of dividends on which the two differ (must be 0). */
long orc_check_canon_division(long n, unsigned long long seed) {
    const double R = 2147483646.0, C = R * R, rc = 1.0 / C;
    unsigned long long s = seed ? seed : 88172645463325252ull;
    long bad = 0;
    for (long i = 0; i < n; i++) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        uint32_t a = (uint32_t) (s % 2147483646u);
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        uint32_t b = (uint32_t) (s % 2147483646u);
        if (i < n / 16) b = (uint32_t) (i & 1023);
        else if (i < n / 8) b = 2147483645u - (uint32_t) (i & 1023);
        double sum = (double) a;
        sum += (double) b * R;
        const double want = sum / C;
        const double q = sum * rc;
        const double got = fma(fma(-q, C, sum), rc, q);
        bad += want != got;
    }
    return bad;
}

static float normal_float_compat(orc_rng *r, float mean, float sd) {
    float x, y, r2;
    do {
        x = 2.0f * canon_float_minstd(r) - 1.0f;
        y = 2.0f * canon_float_minstd(r) - 1.0f;
        r2 = x * x + y * y;
    } while (r2 > 1.0f || r2 == 0.0f);
    float mult = sqrtf(-2 * logf(r2) / r2);
    float ret = y * mult;
    return ret * sd + mean;
}

/* PHILOX mode: standard normal from one 32-bit uniform by two-level tabulated inverse CDF (tools/gen_tables.py):
 * 4096 cells, linear interpolation inside a cell; the lowest 16 cells (p < 2^-8, where the quantile function bends most) are
 * refined by a second 4096-entry table with cells of 2^-20, the highest 16 are their mirror image.  Kolmogorov distance to the
 * normal law <= 2e-6 (tests/test_law_fidelity_cpu.py). */
#define ORC_NORMAL_TAIL_CELLS 16u
static float icdf_interp(const float *tab, uint32_t idx, float frac) {
    float a = tab[idx], b = tab[idx + 1];
    float diff = b - a;
    float prod = diff * frac;
    return a + prod;
}
float orc_normal_from_word(const orc_tables *t, uint32_t w) {
    uint32_t cell = w >> 20;
    if (cell >= 4096u - ORC_NORMAL_TAIL_CELLS) {
        uint32_t m = ~w; /* < 2^24 */
        return -icdf_interp(t->normal_tail, m >> 12, (float) (m & 0xFFFu) * (1.0f / 4096.0f));
    }
    if (cell < ORC_NORMAL_TAIL_CELLS) return icdf_interp(t->normal_tail, w >> 12, (float) (w & 0xFFFu) * (1.0f / 4096.0f));
    return icdf_interp(t->normal_icdf, cell, (float) (w & 0xFFFFFu) * (1.0f / 1048576.0f));
}

/* CHS:804-814 CarArriveRandom::mk_soc, given the N(7,3) draw already narrowed to float */
static float soc_from_experience(float driver_experience) {
    if (driver_experience < 1.0) driver_experience = 1.0;
    else if (driver_experience > 10.0) driver_experience = 10.0;
    float soc = 75.0 - 5.0 * driver_experience;
    return soc;
}
float orc_mk_soc(orc_rng *r) { return soc_from_experience((float) normal_double_compat(r, 7.0, 3.0)); }

/* PHILOX mode: the same variate by inverse-CDF lookup -- 12 bits pick the cell of the tabulated inverse CDF of
 * clip(N(7,3),1,10), 20 bits interpolate linearly inside it (three f32 roundings) */
float orc_soc_from_word(const orc_tables *t, uint32_t w) {
    uint32_t idx = w >> 20;
    float frac = (float) (w & 0xFFFFFu) * (1.0f / 1048576.0f);
    float a = t->soc_d_icdf[idx], b = t->soc_d_icdf[idx + 1];
    float diff = b - a;
    float prod = diff * frac;
    float d = a + prod;
    return soc_from_experience(d);
}

/* PHILOX mode, EV arrivals: the arrival SoC takes one of ORC_SOC_LEVELS = 2048 equiprobable levels (the top 11 bits of
 * the word): level l sits at probability (l + 0.5) / 2048 of the same tabulated inverse CDF, i.e. exactly at its node
 * 2 l + 1.  A finite level set lets an implementation tabulate a car's whole charging history (curve time and power after
 * every number of car_steps) once per level and station. */
float orc_soc_level_value(const orc_tables *t, uint32_t level) { return soc_from_experience(t->soc_d_icdf[2u * level + 1u]); }
float orc_soc_level_from_word(const orc_tables *t, uint32_t w) { return orc_soc_level_value(t, w >> 21); }

/* CHS:816-830 mk_late_time -- both pile classes pass "slow" (CHS:869, CHS:1034): max(0, round(N(2,2))) */
int orc_mk_late_time(orc_rng *r) {
    float car_number = normal_float_compat(r, 2.0f, 2.0f);
    int duration = (int) roundf(car_number);
    if (duration < 0) duration = 0;
    return duration;
}

/* PHILOX mode: the same integer law from one 32-bit uniform and the tabulated CDF */
int orc_late_from_word(const orc_tables *t, uint32_t w) {
    int late = 0;
    for (int j = 0; j < 16; j++) late += (w >= t->late_thr[j]) ? 1 : 0;
    return late;
}

/* CHS:832-842 init_station_car_number(mu, theta=3) */
int orc_init_station_car_number(orc_rng *r, const orc_tables *t, int station, int mu) {
    int theta = 3;
    float car_number;
    if (r->mode == ORC_RNG_COMPAT) {
        car_number = normal_float_compat(r, (float) mu, 1.0f);
    } else {
        uint32_t o[4];
        philox_block(r, ORC_PU_INIT, station, 0, o);
        car_number = orc_normal_from_word(t, o[0]) + (float) mu;
    }
    int temp = (int) roundf(car_number);
    if (temp > mu + theta) temp = mu + theta;
    else if (temp < mu - theta) temp = mu - theta;
    return temp;
}

/* ================================================================================ charge curves */

/* CHS:551-553 */
static float slow_b_part1(float x) {
    return -0.002056 * pow(x, 4) + 0.00921 * pow(x, 3) + 0.03562 * pow(x, 2) + 0.02379 * x + 6.007;
}
/* CHS:555-557 */
static float slow_b_part2(float x) { return (-4.041 * x + 21.1) / (x - 0.485); }
/* CHS:559-562 */
static float slow_c_part1(float x) {
    return -0.0004112 * pow(x, 5) + 0.0023025 * pow(x, 4) + (0.03562 / 3) * pow(x, 3) + 0.011895 * pow(x, 2) +
           6.007 * x;
}
/* CHS:564-566 */
static float slow_c_part2(float x) { return -4.041 * x + 19.140115 * log(fabs(x - 0.485)) + 11.943306312699628; }
/* CHS:568-578 */
static float slow_c_hole(float x) {
    if (x <= 0) return 0;
    else if (x <= 2.33 * 4) return slow_c_part1(x / 4);
    else if (x <= 3.67 * 4) return slow_c_part2(x / 4);
    else return slow_c_part2(3.67);
}
/* CHS:580-582 */
static float slow_s_t_t_1(float soc) { return (-4.276 * pow(10, -5)) * pow(soc, 2) + 0.1295 * soc; }
/* CHS:584-588 */
static float slow_s_t_t_2(float soc) {
    return (4.742 * pow(10, -6)) * pow(soc, 4) - 0.001529 * pow(soc, 3) + 0.1871 * pow(soc, 2) - 10.15 * soc +
           213.1 + 0.1787983924863248;
}
/* CHS:495-511 */
static float slow_time_to_power(float charge_time, int constant_power) {
    if (constant_power) {
        if (14.68 >= charge_time && charge_time >= 0) return 5.254973139368931;
        else return 0;
    } else {
        if (charge_time < 2.33 * 4) return slow_b_part1(charge_time / 4);
        else if (charge_time < 3.67 * 4) return slow_b_part2(charge_time / 4);
        else return 0;
    }
}
/* CHS:513-525 */
static float slow_time_to_soc(float charge_time, int constant_power) {
    if (constant_power) {
        if (charge_time <= 0) return 0;
        else if (charge_time >= 14.68) return 100;
        else return 100 * charge_time / 14.68;
    } else {
        return 100 * slow_c_hole(charge_time) / 19.285746346634653;
    }
}
/* CHS:527-548 */
static float slow_soc_to_time(float soc, int constant_power) {
    if (constant_power) {
        if (soc <= 0) return 0;
        else if (soc >= 100) return 14.68;
        else return 14.68 * soc / 100;
    } else {
        if (soc < 0) return 0;
        else if (0 <= soc && soc <= 73.89239629561729) return slow_s_t_t_1(soc);
        else if (73.89239629561729 < soc && soc <= 100)
            return slow_s_t_t_2(soc) + 0.2012016075138625 * (soc - 73.89239629561729) / 26.10760370438271;
        else return 14.68;
    }
}

/* CHS:684-686 */
static float fast_a_part1(float x) { return 0.7194 * exp(0.053 * x) + 47.78; }
/* CHS:688-690 */
static float fast_a_part2(float x) {
    return 0.0002253 * pow(x, 4) - 0.03572 * pow(x, 3) + 2.016 * pow(x, 2) - 48.76 * x + 457.7;
}
/* CHS:692-695 */
static float fast_aa_part1(float x) {
    float aa_part1_c = (0.7194 / 0.053) * exp(0);
    return (0.7194 / 0.053) * exp(0.053 * x) + 50.15 * x - aa_part1_c;
}
/* CHS:697-704 */
static float fast_aa_part2(float x) {
    float aa_part2_c = fast_aa_part1(28.7) -
                       ((0.0002253 / 5) * pow(28.7, 5) - (0.03572 / 4) * pow(28.7, 4) + (2.016 / 3) * pow(28.7, 3) -
                        (48.76 / 2) * pow(28.7, 2) + 457.7 * 28.7);
    return (0.0002253 / 5) * pow(x, 5) - (0.03572 / 4) * pow(x, 4) + (2.016 / 3) * pow(x, 3) -
           (48.76 / 2) * pow(x, 2) + 457.7 * x + aa_part2_c;
}
/* CHS:706-719 */
static float fast_matlab_fitted_curve(float x) {
    float p1 = -18.18, p2 = 9.559, p3 = 48.99, p4 = -62.97, p5 = 29.09;
    float q1 = -23.9, q2 = 56.48, q3 = -50.12, q4 = 18.96;
    float temp = (p1 * pow(x, 4) + p2 * pow(x, 3) + p3 * pow(x, 2) + p4 * x + p5) /
                 (pow(x, 4) + q1 * pow(x, 3) + q2 * pow(x, 2) + q3 * x + q4);
    return temp;
}
/* CHS:721-725 */
static float fast_norm_soc(float x) {
    float mean = 61.43;
    float std = 31.48;
    return (x - mean) / std;
}
/* CHS:621-637 */
static float fast_time_to_power(float charge_time, int constant_power) {
    if (constant_power) {
        if (3.4133333333333336 >= charge_time && charge_time >= 0) return 36.44764034125146;
        else return 0;
    } else {
        if (charge_time >= 0 && charge_time < (28.7 / 15)) return fast_a_part1(charge_time * 15);
        else if (charge_time >= 0 && charge_time < (51.2 / 15)) return fast_a_part2(charge_time * 15);
        else return 0;
    }
}
/* CHS:639-659 */
static float fast_time_to_soc(float charge_time, int constant_power) {
    if (constant_power) {
        if (charge_time <= 0) return 0;
        else if (charge_time >= 3.4133333333333336) return 100;
        else return 100 * charge_time / 3.4133333333333336;
    } else {
        if (charge_time <= 0) return 0;
        else if (charge_time <= 28.7 / 15) return fast_aa_part1(charge_time * 15) * (100 / fast_aa_part2(51.2));
        else if (charge_time <= 51.2 / 15) return fast_aa_part2(charge_time * 15) * (100 / fast_aa_part2(51.2));
        else return 100;
    }
}
/* CHS:661-681 */
static float fast_soc_to_time(float soc, int constant_power) {
    if (constant_power) {
        if (soc <= 0) return 0;
        else if (soc >= 100) return 3.4133333333333336;
        else return 3.4133333333333336 * soc / 100;
    } else {
        if (soc <= 0) return 0;
        else if (soc >= 100) return 51.2 / 15;
        else return fast_matlab_fitted_curve(fast_norm_soc(soc));
    }
}

float orc_curve_slow(int which, float x, int cp) {
    if (which == 0) return slow_time_to_power(x, cp);
    if (which == 1) return slow_time_to_soc(x, cp);
    return slow_soc_to_time(x, cp);
}
float orc_curve_fast(int which, float x, int cp) {
    if (which == 0) return fast_time_to_power(x, cp);
    if (which == 1) return fast_time_to_soc(x, cp);
    return fast_soc_to_time(x, cp);
}

static float time_to_power(int type, float t, int cp) {
    return type == ORC_FAST ? fast_time_to_power(t, cp) : slow_time_to_power(t, cp);
}
static float time_to_soc(int type, float t, int cp) {
    return type == ORC_FAST ? fast_time_to_soc(t, cp) : slow_time_to_soc(t, cp);
}
static float soc_to_time(int type, float s, int cp) {
    return type == ORC_FAST ? fast_soc_to_time(s, cp) : slow_soc_to_time(s, cp);
}

/* ===================================================================================== station */

/* CHS:204-231 make_init_list + CHS:265-274 pl_reset */
static void station_clear(orc_station *s) {
    for (int i = 0; i < s->n; i++) {
        s->car[i] = s->charge[i] = s->emergency[i] = s->assign[i] = 0;
        s->power[i] = s->soc[i] = s->init_soc[i] = s->target_soc[i] = 0;
        s->p_arrive_soc[i] = s->p_target_soc[i] = s->p_current_soc[i] = s->p_current_power[i] = -1;
        s->stay_time[i] = s->already[i] = -1;
    }
}

/* CHS:1128-1167 / 1438-1477 constructors (the evs_reset they run is left to orc_station_reset) */
void orc_station_init(orc_station *s, int type, int piles, int wait, int constant_charging, int index,
                      int slot_base) {
    memset(s, 0, sizeof *s);
    s->type = type;
    s->n = piles;
    s->wait = wait;
    s->constant_charging = constant_charging;
    s->index = index;
    s->slot_base = slot_base;
    s->constant_power = type == ORC_FAST ? 36.44764034125146 : 5.254973139368931;
    s->transformer_limit = s->constant_power * s->n;
    station_clear(s);
}

/* CHS:276-286 reset_position */
static void reset_position(orc_station *s, int i) {
    s->car[i] = s->charge[i] = s->emergency[i] = s->assign[i] = 0;
    s->power[i] = s->soc[i] = s->init_soc[i] = s->target_soc[i] = 0;
    s->p_arrive_soc[i] = s->p_target_soc[i] = s->p_current_soc[i] = s->p_current_power[i] = -1;
    s->stay_time[i] = s->already[i] = -1;
}

/* CHS:879-898 / 1044-1063 calculate_needed */
static void calculate_needed(orc_station *s, int i) {
    int cc = s->constant_charging;
    float needed_time = soc_to_time(s->type, s->p_target_soc[i], cc) - soc_to_time(s->type, s->p_current_soc[i], cc);
    int time_left = s->stay_time[i] - s->already[i];
    if (needed_time > 0) {
        if (time_left <= ceilf(needed_time)) {
            s->emergency[i] = 10;
        } else {
            s->emergency[i] = pow((needed_time / (float) time_left), 2);
        }
    } else {
        s->emergency[i] = 0.0;
    }
    s->soc[i] = s->p_current_soc[i];
}

/* CHS:1233-1261 / 1544-1572 calculate_output */
static void calculate_output(orc_station *s) {
    for (int i = 0; i < s->n; i++)
        if (s->car[i] > 0.5) calculate_needed(s, i);
    float min_power = 0, max_power = 0, now_power = 0;
    int number = 0;
    if (s->exact_sums) {
        /* production semantics (PHILOX mode): the three sums are order-independent -- every slot power is truncated to
         * a multiple of 2^-19 kW (C cast of power * 2^19 to an integer: |power| < 64 kW), the integers are added exactly (64
         * bits: up to 256 slots), and the total is rounded once to f32.  Within n * 2^-19 kW of the
         * exact sum; the reference's sequential f32 sum (CHS:1244-1255) is within n * 2^-24 relative of it.  An
         * implementation may add the terms in any order and with any grouping (wave butterflies, LDS atomics). */
        int64_t amin = 0, amax = 0, anow = 0;
        for (int i = 0; i < s->n; i++) {
            if (!(s->car[i] > 0.5)) continue;
            number += 1;
            int32_t q = (int32_t) rintf(s->power[i] * 524288.0f); /* exact product, nearest integer (ties to even): kw_to_fixed */
            amax += q;
            if (s->emergency[i] > 8) amin += q;
            if (s->charge[i] <= 1.1 && s->charge[i] >= 0.9) anow += q;
        }
        s->min_power = (float) amin * (1.0f / 524288.0f);
        s->max_power = (float) amax * (1.0f / 524288.0f);
        s->charge_power = (float) anow * (1.0f / 524288.0f);
        s->car_number = number;
        return;
    }
    for (int i = 0; i < s->n; i++) {
        if (s->car[i] > 0.5) {
            number += 1;
            max_power += s->power[i];
            if (s->emergency[i] > 8) min_power += s->power[i];
            if (s->charge[i] <= 1.1 && s->charge[i] >= 0.9) now_power += s->power[i];
        }
    }
    s->min_power = min_power;
    s->max_power = max_power;
    s->charge_power = now_power;
    s->car_number = number;
}

/* CHS:900-909 / 1065-1074 car_step */
static void car_step(orc_station *s, int i) {
    int cc = s->constant_charging;
    float temp_time = soc_to_time(s->type, s->p_current_soc[i], cc) + 1;
    s->p_current_soc[i] = time_to_soc(s->type, temp_time, cc);
    s->p_current_power[i] = time_to_power(s->type, temp_time, cc);
    s->power[i] = time_to_power(s->type, temp_time, cc);
    for (int j = 0; j < s->n; j++) s->assign[j] = 0;
}

/* CHS:912-930 / 1077-1095 remove_car (quick_leave is hard-wired false, CHS:1121,1431) */
static void remove_car(orc_station *s, int i) {
    s->already[i] += 1;
    int time_left = s->stay_time[i] - s->already[i];
    if (time_left <= 0) reset_position(s, i);
}

/* CHS:864-877 / 1029-1042 add_car (+ calculate_min_charging_time CHS:933-937 / 1098-1102) */
static void add_car(orc_station *s, orc_rng *r, const orc_tables *t, int i) {
    int cc = s->constant_charging;
    int slot = s->slot_base + i;
    int late;
    if (r->mode == ORC_RNG_COMPAT) {
        s->p_arrive_soc[i] = orc_mk_soc(r);
        s->p_current_soc[i] = s->p_arrive_soc[i];
        s->p_target_soc[i] = orc_uniform_level((int) (orc_glibc_rand(r) % 1000u), 80, 100);
    } else {
        /* PHILOX: one block per admitted slot -- word 0 arrival SoC, word 1 target level, word 2 extra stay */
        uint32_t o[4];
        philox_block(r, ORC_PU_SOC, slot, 0, o);
        s->p_arrive_soc[i] = orc_soc_level_from_word(t, o[0]);
        s->p_current_soc[i] = s->p_arrive_soc[i];
        s->p_target_soc[i] = orc_uniform_level((int) (o[1] % 1000u), 80, 100);
        late = orc_late_from_word(t, o[2]);
    }
    float needed_time = soc_to_time(s->type, s->p_target_soc[i], cc) - soc_to_time(s->type, s->p_current_soc[i], cc);
    int must_needed = (int) ceilf(needed_time);
    if (r->mode == ORC_RNG_COMPAT) late = orc_mk_late_time(r);
    s->stay_time[i] = must_needed + late;
    s->already[i] = 0;
    s->car[i] = 1; /* occupy_position CHS:314-317 */
    s->assign[i] = 0;
    s->power[i] = time_to_power(s->type, soc_to_time(s->type, s->p_current_soc[i], cc), cc);
    s->init_soc[i] = s->p_arrive_soc[i];
    s->target_soc[i] = s->p_target_soc[i];
}

/* CHS:442-451 find_empty */
static void find_empty(orc_station *s) {
    s->empty_number = 0;
    for (int i = 0; i < s->n; i++)
        if (s->car[i] <= 0.5) s->empty_list[s->empty_number++] = i;
}

/* CHS:417-430 assign_car */
static void assign_car(orc_station *s) {
    int assign_number;
    if (s->wait) {
        assign_number = (s->line + s->flow_in_last) < s->empty_number ? (s->line + s->flow_in_last) : s->empty_number;
        s->line = s->line + s->flow_in_last - assign_number;
        s->line = s->line < 10 ? s->line : 10; /* max_line CHS:197 */
    } else {
        assign_number = s->flow_in_last < s->empty_number ? s->flow_in_last : s->empty_number;
    }
    for (int i = 0; i < assign_number; i++) s->assign[s->empty_list[i]] = 1;
}

/* CHS:1272-1316 (slow) / 1583-1627 (fast) receive_car */
static void receive_car(orc_station *s, orc_rng *r, const orc_tables *t, int reset_evs) {
    find_empty(s); /* tell_empty CHS:401-407 */
    int in_car;
    if (reset_evs) {
        in_car = orc_init_station_car_number(r, t, s->index, (int) round(s->n / 2));
    } else {
        int in_car_time = s->time_hole % 96;
        int n = orc_arrival_index(t, in_car_time, orc_draw_k(r, ORC_PU_ARRIVE, s->index, 0));
        in_car = s->type == ORC_FAST ? orc_count_fast(n) : orc_count_slow(n);
    }
    float a;
    float true_line = 0;
    float leave_possibility;
    for (int wait_id = 0; wait_id < s->line; wait_id++) {
        leave_possibility = 0.1 * logf(wait_id + 1);
        a = orc_uniform_level(orc_draw_k(r, ORC_PU_RENEGE, s->index, wait_id), 0, 1);
        if (a > leave_possibility) true_line += 1;
    }
    s->line = true_line;
    int true_in_car = 0;
    float arrival_stay_possibility;
    for (int arrive_id = 0; arrive_id < in_car; arrive_id++) {
        a = orc_uniform_level(orc_draw_k(r, ORC_PU_ARRIVE, s->index, 1 + arrive_id), 0, 1); /* philox: word 1+j */
        arrival_stay_possibility = expf(-(0.01 * (s->line + arrive_id)));
        if (a <= arrival_stay_possibility && arrive_id <= s->n) true_in_car += 1;
    }
    /* slow records the thinned count (CHS:1306), fast the raw one (CHS:1617) */
    s->flow_in_last = s->type == ORC_FAST ? in_car : true_in_car;
    s->has_flow = 1;
    assign_car(s);
    for (int i = 0; i < s->n; i++)
        if (s->assign[i] > 0.5) add_car(s, r, t, i);
    for (int i = 0; i < s->n; i++) s->assign[i] = 0;
}

/* CHS:1209-1231 / 1520-1542 evs_reset */
void orc_station_reset(orc_station *s, orc_rng *r, const orc_tables *t) {
    s->has_flow = 0;
    s->flow_in_last = 0;
    s->line = 0;
    s->time_hole = 0;
    s->load_assigned = 0;
    s->empty_number = 0;
    s->min_power = s->max_power = s->charge_power = 0;
    s->car_number = 0;
    station_clear(s);
    receive_car(s, r, t, 1);
    calculate_output(s);
}

/* CHS:1404-1413 / 1716-1725 judge_feasibility */
static void judge_feasibility(const orc_station *s, float *actions) {
    for (int i = 0; i < s->n; i++) {
        if (s->emergency[i] >= 1.01 && s->car[i] == 1) actions[i] = 1;
        else if (s->car[i] < 1) actions[i] = 0;
    }
}
/* CHS:1364-1373 / 1676-1685 assign_on_off_piece */
static void assign_on_off_piece(orc_station *s, const float *actions) {
    for (int i = 0; i < s->n; i++) s->charge[i] = 0;
    for (int i = 0; i < s->n; i++)
        if (s->car[i] == 1 && actions[i] == 1) s->charge[i] = 1;
}

static void step_tail(orc_station *s, orc_rng *r, const orc_tables *t) {
    for (int i = 0; i < s->n; i++) {
        if (s->charge[i]) car_step(s, i);
        if (s->car[i]) remove_car(s, i);
    }
    receive_car(s, r, t, 0);
    s->time_hole = (s->time_hole + 1) % 96;
    calculate_output(s);
}

/* CHS:1188-1207 / 1499-1518 evs_step(std::vector<float>) */
void orc_station_step(orc_station *s, orc_rng *r, const orc_tables *t, const float *actions_in) {
    float actions[ORC_MAX_PILES];
    memcpy(actions, actions_in, sizeof(float) * (size_t) s->n);
    calculate_output(s);
    judge_feasibility(s, actions);
    assign_on_off_piece(s, actions);
    step_tail(s, r, t);
}

/* sort of CHS:1324-1336: multimap<float,int> keyed by -emergency => emergency descending, ties by index */
static void emergency_order(const orc_station *s, int *order) {
    for (int i = 0; i < s->n; i++) order[i] = i;
    for (int i = 1; i < s->n; i++) { /* stable insertion sort */
        int v = order[i];
        float kv = -s->emergency[v];
        int j = i - 1;
        while (j >= 0 && -s->emergency[order[j]] > kv) { order[j + 1] = order[j]; j--; }
        order[j + 1] = v;
    }
}

/* CHS:1318-1362 / 1629-1674 assign_on_off (scalar-load mode) */
static void assign_on_off(orc_station *s) {
    int order[ORC_MAX_PILES];
    for (int i = 0; i < s->n; i++) s->charge[i] = 0;
    emergency_order(s, order);
    if (s->constant_charging) {
        int constant_charge_number = (int) roundf(s->load_assigned / s->constant_power);
        int assigned = 0;
        for (int h = 0; h < s->n; h++) {
            if (s->car[order[h]] == 1 && assigned < constant_charge_number) {
                s->charge[order[h]] = 1;
                assigned += 1;
            }
        }
    } else {
        /* rank_power_add CHS:1375-1402 / 1687-1714 */
        float added_temp = 0;
        for (int h = 0; h < s->n; h++) {
            if (s->car[order[h]] == 1) {
                added_temp += s->power[order[h]];
                if (s->load_assigned + 0.0001 >= added_temp) s->charge[order[h]] = 1;
            }
        }
    }
}

/* CHS:1169-1186 / 1480-1497 evs_step(float) with catch_load CHS:358-366 */
void orc_station_step_load(orc_station *s, orc_rng *r, const orc_tables *t, float load) {
    calculate_output(s);
    if (load > s->max_power) load = s->max_power;
    else if (load < s->min_power) load = s->min_power;
    s->load_assigned = load;
    assign_on_off(s);
    step_tail(s, r, t);
}

/* ==================================================================================== hydrogen */

/* HYD:338-388 __init_to_target_pressure on lookup_table HYD:232-233 */
double orc_j2601_target_pressure(double pressure) {
    static const double X[10] = {0.50, 5.00, 10.0, 15.0, 20.0, 30.0, 40.0, 50.0, 60.0, 70.0};
    static const double Y[10] = {87.4, 81.0, 86.8, 86.1, 85.4, 83.8, 82.2, 80.4, 78.5, 76.1};
    if (pressure < X[1]) return Y[0];
    for (int a = 0; a < 8; a++) {
        int in = (a < 7) ? (X[1 + a] <= pressure && pressure < X[2 + a]) : (X[1 + a] <= pressure && pressure <= X[2 + a]);
        if (in) return (Y[2 + a] - Y[1 + a]) / (X[2 + a] - X[1 + a]) * (pressure - X[1 + a]) + Y[1 + a];
    }
    return pressure;
}

/* HYD:390-391 */
static double pressure_to_mass(double pressure) { return (6.3 * 1000) * (pressure / 70); }

/* HYD:308-321 (+ ramp rate HYD:330-336) */
void orc_j2601_time_mass(double p0, double *time_need, double *mass_need) {
    const double APRR_top_off = 7.2, APRR_norm_val = 18.5;
    if (p0 < 5) {
        *time_need = (69 - p0) / APRR_norm_val + (87.4 - 69) / APRR_top_off;
        double target = orc_j2601_target_pressure(p0);
        *mass_need = pressure_to_mass(target) - pressure_to_mass(p0);
        return;
    }
    double target = orc_j2601_target_pressure(p0);
    double aprr = (p0 < 5) ? APRR_top_off : ((5 <= p0 && p0 < 70) ? APRR_norm_val : 0);
    *time_need = (target - p0) / aprr;
    *mass_need = pressure_to_mass(target) - pressure_to_mass(p0);
}

/* HYD:10-24 Electrolyser.__init__: cells = ceil(mass_flow_max / g/s of one cell at 10 mL/min) */
long orc_electrolyser_cells(double mass_flow_max) {
    double v_M = 0.082 * (273 + 25) / 1;
    double v_H = 10;
    double v_H_L = (v_H / 1000) / 60;
    double v_H_mol = v_H_L / v_M;
    double v_H_mass = v_H_mol * 2.02;
    return (long) ceil(mass_flow_max / v_H_mass);
}

/* HYD:38-48 Electrolyser.get_power */
double orc_electrolyser_power(double v_H_mass_in, long cells) {
    if (cells == 0) return 0;
    double v_M = 0.082 * (273 + 25) / 1;
    double v_H_mass = v_H_mass_in / cells;
    double v_H_mol = v_H_mass / 2.02;
    double v_H_L = v_H_mol * v_M;
    double v_H = v_H_L * 1000 * 60;
    double temp = v_H * 2 * 96487 / (v_M * 1000 * 60);
    double power = pow(temp, 2) * 0.326 + temp * 1.476;
    power = cells * power / 1000;
    return power;
}

/* HYD:57-82 Compressor */
double orc_compressor_kw(double m_H2) {
    double eta_c = 0.8, alpha = 1.4, R = 0.082, T = 273 + 25, P_in = 1, P_out = 200;
    double P_a = sqrt(P_in * P_out);
    double part1 = alpha / (alpha - 1);
    double part2 = part1 * R * T;
    double part3 = (alpha - 1) / alpha;
    double n_H2 = m_H2 / 2.02;
    double W_1 = part2 * (-1 + pow(P_a / P_in, part3));
    double W_2 = part2 * (-1 + pow(P_out / P_a, part3));
    double W_c = n_H2 * (W_1 + W_2) / eta_c;
    return W_c / 1000;
}

/* HYD:253-285 hvs_step */
static void hvs_step(orc_env *e, int time) {
    const double time_interval = 15;
    e->hv_num = 0; /* _hvs_interval_reset HYD:287-292 */
    int k = orc_draw_k(&e->rng, ORC_PU_HV, 0, 0);
    int n = orc_arrival_index(e->tab, time, k);
    e->hv_arrive = orc_count_hv(n, (float) 0.3, (float) e->cfg.fcev_permeate); /* HYD:247-251 */
    for (int j = 0; j < e->hv_arrive; j++) {
        double soc;
        if (e->rng.mode == ORC_RNG_COMPAT) {
            soc = (double) orc_mk_soc(&e->rng);
        } else {
            uint32_t o[4];
            philox_block(&e->rng, ORC_PU_HVSOC, j, 0, o);
            soc = (double) orc_soc_from_word(e->tab, o[0]);
        }
        if (soc < 0.5) soc = 0.5;
        double p0 = (soc * 0.01) * 70; /* _soc_to_pressure HYD:302-306 */
        double tn, mn;
        orc_j2601_time_mass(p0, &tn, &mn);
        if (e->q_len < ORC_QCAP) {
            e->q_time[e->q_len] = tn;
            e->q_mass[e->q_len] = mn;
            e->q_len++;
        } else {
            e->q_overflow = 1;
        }
    }
    double total_time = 0;
    for (int i = 0; i < e->q_len; i++) total_time += e->q_time[i];
    double total_mass = 0;
    for (int i = 0; i < e->q_len; i++) total_mass += e->q_mass[i];
    if (total_time > time_interval) {
        for (int i = 1; i <= e->hv_arrive - 1; i++) {
            double part = 0;
            int keep = e->q_len - i;
            if (keep < 0) keep = 0;
            for (int j = 0; j < keep; j++) part += e->q_time[j];
            if (part <= time_interval) {
                e->hv_line = i;
                e->hv_num = keep;
                break;
            }
        }
        e->total_mass_need = total_mass;
        int num = e->hv_num;
        for (int j = num; j < e->q_len; j++) {
            e->q_time[j - num] = e->q_time[j];
            e->q_mass[j - num] = e->q_mass[j];
        }
        e->q_len -= num;
    } else {
        e->total_mass_need = total_mass;
        e->hv_line = 0;
        e->hv_num = e->q_len;
        e->q_len = 0;
    }
}

/* HYD:104-126 sty_step */
static void sty_step(orc_env *e, double mass_s_H2, double hy_use) {
    double temp_to_store = mass_s_H2 * 15 * 60;
    e->capacity += temp_to_store;
    double lower_change = e->capacity - 0.1 * e->cap_mass;
    lower_change = lower_change > 0 ? lower_change : 0;
    double temp_to_change = hy_use < lower_change ? hy_use : lower_change;
    e->hy_use = temp_to_change;
    e->not_meet = hy_use - temp_to_change;
    e->capacity -= temp_to_change;
    e->capacity -= e->capacity * e->cfg.hydro_loss;
    e->store_soc = e->capacity / e->cap_mass;
}

/* HYD:160-195 hy_step; demand == NULL -> run hvs_step, else use *demand (construction sweep) */
static double hy_step(orc_env *e, double gen_speed, const double *demand) {
    if (demand) e->total_mass_need = *demand;
    else hvs_step(e, e->sys_time);
    double must_charge = e->cap_mass * 0.1 - e->capacity;
    must_charge = must_charge > 0 ? must_charge : 0;
    double upper_charge = e->cap_mass - e->capacity;
    upper_charge = upper_charge > 0 ? upper_charge : 0;
    double charge_temp = gen_speed * e->v_h_max * (15 * 60);
    charge_temp = charge_temp < upper_charge ? charge_temp : upper_charge;
    charge_temp = charge_temp > must_charge ? charge_temp : must_charge;
    e->hy_flow_speed = charge_temp / (15 * 60);
    e->hy_flow_speed = e->hy_flow_speed < e->v_h_max ? e->hy_flow_speed : e->v_h_max;
    e->ele_power = orc_electrolyser_power(e->hy_flow_speed, e->cell_number);
    e->cpr_power = orc_compressor_kw(e->hy_flow_speed);
    sty_step(e, e->hy_flow_speed, e->total_mass_need);
    e->all_power_second = e->ele_power + e->cpr_power;
    e->sys_time = (e->sys_time + 1) % 96;
    return e->all_power_second;
}

/* HYD:197-208 hy_reset */
static void hy_reset(orc_env *e) {
    e->q_len = 0;
    e->hv_line = 0;
    e->hv_num = 0;
    e->capacity = e->cfg.init_soc * e->cap_mass;
    e->store_soc = e->cfg.init_soc;
    e->sys_time = 0;
}

/* HYD:409-430 HFC.use_cell */
static double use_cell(orc_env *e, double fc_power, double charging_p) {
    double cell_number = e->cfg.fc_max_power;
    if (fc_power > cell_number) fc_power = cell_number;
    else if (fc_power < 0) fc_power = 0;
    else if (fc_power > charging_p) fc_power = charging_p;
    double hy_to_use = fc_power * 1500 / 119.6;
    hy_to_use = e->capacity < hy_to_use ? e->capacity : hy_to_use; /* min(capacity, hy_to_use) */
    e->hy_to_use = hy_to_use;
    double true_power = fc_power > 0 ? fc_power : 0; /* HYD:426 overrides HYD:425 */
    e->capacity -= hy_to_use;
    return true_power;
}

/* ==================================================================================== env host */

/* numpy pairwise sum for n <= 128, n % 8 == 0 (np.mean / np.std at MGR:45-46) */
static double np_sum96(const double *a) {
    double r[8];
    for (int j = 0; j < 8; j++) r[j] = a[j];
    for (int i = 8; i < 96; i += 8)
        for (int j = 0; j < 8; j++) r[j] += a[i + j];
    return ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
}

int orc_env_obs_dim(const orc_config *cfg) {
    int active = (cfg->piles[0] > 0) + (cfg->piles[1] > 0);
    return 2 + 4 * active + 3;
}

/* MGR:25-130 __init__ (+ AGG:39-90, HYD:134-158, HYD:394-406, REN:22-36) */
void orc_env_init(orc_env *e, const orc_config *cfg, const orc_tables *t) {
    memset(e, 0, sizeof *e);
    e->cfg = *cfg;
    e->tab = t;
    orc_station_init(&e->st[0], cfg->type[0], cfg->piles[0], 1, cfg->constant_charging, 0, 0);
    orc_station_init(&e->st[1], cfg->type[1], cfg->piles[1], 1, cfg->constant_charging, 1, cfg->piles[0]);
    double density = 0.089 * (200 / 1);                  /* HYD:94 */
    double capacity_temp = cfg->hydro_store_vlt * 1000;  /* HYD:97 */
    e->cap_mass = density * capacity_temp;               /* HYD:98 */
    e->v_h_max = 0.089 * cfg->hydro_prod_rate * 1000 / 3600; /* HYD:144 */
    e->cell_number = orc_electrolyser_cells(e->v_h_max);
    e->cpr_kw_per_gs = orc_compressor_kw(1.0);
    /* HYD:154-158 action->power table.  The reference runs 101 real hy_step()s here, including random
     * FCEV demand drawn from the process-global streams; the oracle sweeps with zero demand, which is
     * identical whenever the tank clamps (HYD:172-173) do not bind during the sweep. */
    e->capacity = cfg->init_soc * e->cap_mass;
    e->store_soc = cfg->init_soc;
    const double zero = 0.0;
    for (int i = 0; i < 101; i++) {
        e->hy_table[i] = hy_step(e, 0.01 * i, &zero);
        e->hy_table_in[i] = 0.01 * i;
    }
    e->hy_table[101] = e->hy_table[100];
    e->hy_table_in[101] = e->hy_table_in[100];
    hy_reset(e);
    double mean = np_sum96(t->price) / 96; /* MGR:45 */
    double dev[96];
    for (int i = 0; i < 96; i++) {
        double d = t->price[i] - mean;
        dev[i] = fabs(d) * fabs(d);
    }
    e->price_mean = mean;
    e->price_std = sqrt(np_sum96(dev) / 96); /* MGR:46 */
    e->ou_pv = e->ou_wd = e->ou_price = 0;
    e->price_count = 0;
    e->obs_dim = orc_env_obs_dim(cfg);
}

/* REN:71-76 OU_Noise.sample */
static double ou_sample(double *state, double theta, double sigma, double z) {
    double dx = theta * (0.0 - *state) + sigma * z;
    *state += dx;
    return *state;
}

/* MGR:344-373 make_state + MGR:318-342 state_norm */
static void make_state(orc_env *e, int time, const double *exo_z, double *obs) {
    const double scale_pv = 5, scale_wd = 1;
    double z[3];
    uint32_t ow[4] = {0, 0, 0, 0};
    if (!exo_z) philox_block(&e->rng, ORC_PU_OU, 0, 0, ow); /* PHILOX: word 0 pv, 1 wind, 2 price */
    /* REN:38-49 */
    double temp = e->tab->pv[e->pv_day][time];
    if (temp > 0 && e->pv_day % 2 == 0) {
        z[0] = exo_z ? exo_z[0] : (double) orc_normal_from_word(e->tab, ow[0]);
        temp += ou_sample(&e->ou_pv, .01, 1., z[0]) * (1 + e->cfg.renew_fluctuate);
    }
    e->re_pv = (temp > 0 ? temp : 0) * scale_pv;
    temp = e->tab->wd[e->wd_day][time];
    z[1] = exo_z ? exo_z[1] : (double) orc_normal_from_word(e->tab, ow[1]);
    temp += ou_sample(&e->ou_wd, .01, 1.5, z[1]) * (1 + e->cfg.renew_fluctuate);
    e->re_wd = (temp > 0 ? temp : 0) * scale_wd;
    /* MGR:354-360 */
    double price_next;
    if (e->price_count % 4 == 0) {
        z[2] = exo_z ? exo_z[2] : (double) orc_normal_from_word(e->tab, ow[2]);
        price_next = ou_sample(&e->ou_price, .1, 0.005, z[2]) * (1 + e->cfg.price_fluctuate);
        e->price_noise_part = price_next;
        price_next += e->price_last;
    } else {
        price_next = e->price_last + e->price_noise_part;
    }
    e->price_count += 1;
    e->time_now = time;
    e->price_next = price_next;
    /* MGR:318-342 */
    int o = 0;
    double k = 2 * M_PI / 96;
    obs[o++] = sin(k * (double) time);
    obs[o++] = (price_next - e->price_mean) / e->price_std;
    for (int s = 0; s < 2; s++) {
        const orc_station *st = &e->st[s];
        if (st->n > 0) {
            double half_range = (double) st->transformer_limit / 2;
            obs[o++] = ((double) st->min_power - half_range) / half_range;
            obs[o++] = ((double) st->charge_power - half_range) / half_range;
            obs[o++] = ((double) st->max_power - half_range) / half_range;
            obs[o++] = (double) st->line / 5;
        }
    }
    obs[o++] = e->store_soc;
    obs[o++] = e->re_pv / (42 * scale_pv);
    obs[o++] = e->re_wd / (92 * scale_wd);
}

/* MGR:304-316 reset (+ REN:51-53, AGG:157-175, HYD:197-208) */
void orc_env_reset(orc_env *e, const int *exo_days, const double *exo_z, double *obs) {
    e->rng.tick += 1;
    if (exo_days) {
        e->pv_day = exo_days[0];
        e->wd_day = exo_days[1];
    } else {
        uint32_t o[4];
        philox_block(&e->rng, ORC_PU_DAY, 0, 0, o);
        e->pv_day = (int) (o[0] % 100u);
        e->wd_day = (int) (o[1] % 150u);
    }
    orc_station_reset(&e->st[0], &e->rng, e->tab);
    orc_station_reset(&e->st[1], &e->rng, e->tab);
    e->agg_time = 0;
    e->price_last = e->tab->price[95]; /* self.price = [] + mean_for_MAD; price[-1] */
    hy_reset(e);
    make_state(e, 0, exo_z, obs);
    e->price_count = 0;
}

/* MGR:136-302 step */
static void env_step_impl(orc_env *e, const float *action, const double *exo_z, double *obs, double *reward, int *done,
                          int load_mode);

void orc_env_step(orc_env *e, const float *action, const double *exo_z, double *obs, double *reward, int *done) {
    env_step_impl(e, action, exo_z, obs, reward, done, 0);
}

/* Scalar-load control mode: instead of one bit per pile the hub is driven with one kW target per
 * station, evs_step(float) (CHS:1169-1186 / 1480-1497; bound at MAIN:196-197,248-249).  action[0] and
 * action[piles[0]] carry the two loads; the tail entries keep their meaning.  The reference's gym host never takes
 * this path (AGG:141-142 always passes a Vector_float); the rest of step() is unchanged. */
void orc_env_step_load(orc_env *e, const float *action, const double *exo_z, double *obs, double *reward, int *done) {
    env_step_impl(e, action, exo_z, obs, reward, done, 1);
}

static void env_step_impl(orc_env *e, const float *action, const double *exo_z, double *obs, double *reward, int *done,
                          int load_mode) {
    e->rng.tick += 1;
    int S0 = e->cfg.piles[0], S1 = e->cfg.piles[1], S = S0 + S1;
    int time_real_next = (int) e->time_now + 1;
    double re_new_power = e->re_wd + e->re_pv;
    /* MGR:384-404 action_to_real: pile bits in f32, tail swapped */
    float bits[2 * ORC_MAX_PILES];
    for (int i = 0; i < S; i++) {
        float real_action = (action[i] + 1.0f) / 2.0f;
        bits[i] = real_action >= 0.5f ? 1.0f : 0.0f;
    }
    double a_fc = ((double) action[S + 1] + 1) / 2; /* action_real[-2] */
    double a_el = ((double) action[S] + 1) / 2;     /* action_real[-1] */
    /* AGG:116-155 ag_step */
    if (load_mode) {
        orc_station_step_load(&e->st[0], &e->rng, e->tab, action[0]);
        orc_station_step_load(&e->st[1], &e->rng, e->tab, action[S0]);
    } else {
        orc_station_step(&e->st[0], &e->rng, e->tab, bits);
        orc_station_step(&e->st[1], &e->rng, e->tab, bits + S0);
    }
    e->price_last = e->tab->price[e->agg_time];
    e->agg_time = (e->agg_time + 1) % 96;
    double P[2] = {(double) e->st[0].charge_power, (double) e->st[1].charge_power};
    double charging_power = 0 + P[0] + P[1];
    /* MGR:160-180 electrolyser clamp */
    double hy_power_limit = 2000 + re_new_power - charging_power;
    hy_power_limit = hy_power_limit > 0 ? hy_power_limit : 0;
    double act;
    if (e->hy_table[(int) ceil(a_el * 100)] > hy_power_limit) {
        int found = 0;
        act = 0;
        for (int ind = 0; ind < 102; ind++) {
            if (e->hy_table[ind] >= hy_power_limit) {
                act = ind == 0 ? e->hy_table_in[101] : e->hy_table_in[ind - 1]; /* python list[-1] */
                found = 1;
                break;
            }
        }
        if (!found) act = e->hy_table_in[101];
    } else {
        act = a_el;
    }
    hy_step(e, act, NULL);
    e->hy_act = act;
    int gen_hy = e->hy_flow_speed > 0.5;
    /* MGR:183-213 renewable netting */
    double hydrogen_power = e->all_power_second;
    double ev_power_sum = charging_power;
    double ev_list[2] = {P[0], P[1]};
    if (0 + P[0] + P[1] > 0) {
        double fc_rate = charging_power / (0 + P[0] + P[1]);
        ev_list[0] = fc_rate * P[0];
        ev_list[1] = fc_rate * P[1];
    }
    double used_renew = 0;
    if (re_new_power >= hydrogen_power) {
        re_new_power -= hydrogen_power;
        used_renew += hydrogen_power;
        hydrogen_power = 0;
        double tmp = ev_power_sum;
        ev_power_sum = ev_power_sum - re_new_power;
        ev_power_sum = ev_power_sum > 0 ? ev_power_sum : 0;
        used_renew += tmp - ev_power_sum;
        double sl = 0 + ev_list[0] + ev_list[1];
        if (sl > 0) {
            double rate = ev_power_sum / sl;
            ev_list[0] = rate * ev_list[0];
            ev_list[1] = rate * ev_list[1];
        }
    } else {
        hydrogen_power -= re_new_power;
        used_renew = re_new_power;
    }
    e->used_renew = used_renew;
    e->ev_list[0] = ev_list[0]; /* self.re_ev_power_list is taken BEFORE the fuel-cell rescale, MGR:212 */
    e->ev_list[1] = ev_list[1];
    /* MGR:215-227 fuel cell */
    double fc_power;
    if (gen_hy) fc_power = use_cell(e, 0, ev_power_sum);
    else fc_power = use_cell(e, e->cfg.fc_max_power * a_fc, ev_power_sum);
    ev_power_sum -= fc_power;
    e->fc_power = fc_power;
    {
        double sl = 0 + ev_list[0] + ev_list[1];
        if (sl > 0) {
            double rate_ = ev_power_sum / sl;
            ev_list[0] = rate_ * ev_list[0];
            ev_list[1] = rate_ * ev_list[1];
        }
    }
    double hy_loss = -6 / 1000.0 * e->hy_to_use;
    e->hydrogen_power_grid = hydrogen_power;
    e->ev_net[0] = ev_list[0]; /* ... and what the incomes and cumulated_draw_ele use is the list AFTER it, MGR:219-224, 262 */
    e->ev_net[1] = ev_list[1];
    e->ev_sum_net = ev_power_sum; /* self.real_charging_power / self.re_ev_power_sum, MGR:229-231 */
    e->price_now = e->price_next; /* real_state[1] as this step found it, MGR:234 */
    /* MGR:233-269 incomes and reward */
    double real_price_dollar = e->price_next / 4;
    double income_evs_fast = 0.42 / 4 * P[0] - real_price_dollar * ev_list[0];
    double income_evs_slow = 0.21 / 4 * P[1] - real_price_dollar * ev_list[1];
    double income_evs = income_evs_fast + income_evs_slow;
    double income_evs_serve = 0.8 * (0 + e->st[0].flow_in_last + e->st[1].flow_in_last);
    double income_hys = 6 / 1000.0 * e->hy_use;
    double not_meet_loss = -10 / 1000.0 * e->not_meet;
    double hy_cost = -real_price_dollar * hydrogen_power;
    e->income = income_hys + income_evs + income_evs_serve + hy_cost;
    *reward = (income_hys + income_evs + income_evs_serve + hy_cost + 1 * hy_loss + not_meet_loss) / 50;
    e->reward = *reward;
    *done = time_real_next >= 96;
    time_real_next = time_real_next % 96;
    make_state(e, time_real_next, exo_z, obs);
}

/* ================================================================================ vector front */

struct orc_vec {
    long n;
    int obs_dim, act_dim;
    orc_env *envs;
};

long orc_sizeof_env(void) { return (long) sizeof(orc_env); }

orc_vec *orc_vec_create(const orc_config *cfg, const orc_tables *t, long n_envs, long env_id0, int rng_mode,
                        uint64_t seed) {
    orc_vec *v = (orc_vec *) calloc(1, sizeof *v);
    v->n = n_envs;
    v->obs_dim = orc_env_obs_dim(cfg);
    v->act_dim = cfg->piles[0] + cfg->piles[1] + 2;
    v->envs = (orc_env *) malloc(sizeof(orc_env) * (size_t) n_envs);
    for (long i = 0; i < n_envs; i++) {
        orc_env_init(&v->envs[i], cfg, t);
        uint32_t id = (uint32_t) (env_id0 + i);
        if (rng_mode == ORC_RNG_COMPAT) {
            /* per-env seeds of the two reference streams: srand(seed + 2*id + 1), e.seed(seed + 2*id + 2) */
            orc_rng_seed_compat(&v->envs[i].rng, (uint32_t) seed + 2u * id + 1u, (uint32_t) seed + 2u * id + 2u);
        } else {
            orc_rng_seed_philox(&v->envs[i].rng, seed, id);
            v->envs[i].st[0].exact_sums = 1;
            v->envs[i].st[1].exact_sums = 1;
        }
    }
    return v;
}
void orc_vec_destroy(orc_vec *v) {
    if (!v) return;
    free(v->envs);
    free(v);
}
orc_env *orc_vec_env(orc_vec *v, long i) { return &v->envs[i]; }

void orc_vec_reset(orc_vec *v, const int *exo_days, const double *exo_z, double *obs) {
    for (long i = 0; i < v->n; i++)
        orc_env_reset(&v->envs[i], exo_days ? exo_days + 2 * i : NULL, exo_z ? exo_z + 3 * i : NULL,
                      obs + (long) v->obs_dim * i);
}

typedef struct {
    orc_vec *v;
    int load_mode;
    const float *actions;
    const double *exo_z;
    double *obs, *reward;
    unsigned char *done;
    long lo, hi;
} vec_job;

static void *vec_worker(void *p) {
    vec_job *j = (vec_job *) p;
    orc_vec *v = j->v;
    for (long i = j->lo; i < j->hi; i++) {
        int d;
        env_step_impl(&v->envs[i], j->actions + (long) v->act_dim * i, j->exo_z ? j->exo_z + 3 * i : NULL,
                      j->obs + (long) v->obs_dim * i, &j->reward[i], &d, j->load_mode);
        j->done[i] = (unsigned char) d;
    }
    return NULL;
}

static void vec_step_impl(orc_vec *v, const float *actions, const double *exo_z, double *obs, double *reward,
                          unsigned char *done, int n_threads, int load_mode);

void orc_vec_step(orc_vec *v, const float *actions, const double *exo_z, double *obs, double *reward,
                  unsigned char *done, int n_threads) {
    vec_step_impl(v, actions, exo_z, obs, reward, done, n_threads, 0);
}
void orc_vec_step_load(orc_vec *v, const float *actions, const double *exo_z, double *obs, double *reward,
                       unsigned char *done, int n_threads) {
    vec_step_impl(v, actions, exo_z, obs, reward, done, n_threads, 1);
}

static void vec_step_impl(orc_vec *v, const float *actions, const double *exo_z, double *obs, double *reward,
                          unsigned char *done, int n_threads, int load_mode) {
    if (n_threads < 1) n_threads = 1;
    if (n_threads > 256) n_threads = 256;
    if ((long) n_threads > v->n) n_threads = (int) (v->n > 0 ? v->n : 1);
    vec_job jobs[256];
    pthread_t th[256];
    long chunk = (v->n + n_threads - 1) / n_threads;
    for (int k = 0; k < n_threads; k++) {
        long lo = chunk * k, hi = lo + chunk;
        if (hi > v->n) hi = v->n;
        if (lo > hi) lo = hi;
        jobs[k] = (vec_job){v, load_mode, actions, exo_z, obs, reward, done, lo, hi};
    }
    if (n_threads == 1) {
        vec_worker(&jobs[0]);
        return;
    }
    for (int k = 0; k < n_threads; k++) pthread_create(&th[k], NULL, vec_worker, &jobs[k]);
    for (int k = 0; k < n_threads; k++) pthread_join(th[k], NULL);
}

/* ============================================================== accessors for the ctypes tests */

orc_tables *orc_tables_load(const char *data_dir) {
    orc_tables *t = (orc_tables *) calloc(1, sizeof *t);
    char p[4096];
    snprintf(p, sizeof p, "%s/car_flow_possibility_list_save.csv", data_dir);
    if (orc_load_cdf_csv(p, t)) { free(t); return NULL; }
    snprintf(p, sizeof p, "%s/price_96.f64", data_dir);
    if (orc_load_f64(p, t->price, 96)) { free(t); return NULL; }
    snprintf(p, sizeof p, "%s/pv_100x96.f64", data_dir);
    if (orc_load_f64(p, &t->pv[0][0], 100 * 96)) { free(t); return NULL; }
    snprintf(p, sizeof p, "%s/wd_150x96.f64", data_dir);
    if (orc_load_f64(p, &t->wd[0][0], 150 * 96)) { free(t); return NULL; }
    snprintf(p, sizeof p, "%s/soc_d_icdf_4097.f32", data_dir);
    FILE *f = fopen(p, "rb");
    if (!f || fread(t->soc_d_icdf, 4, 4097, f) != 4097) { if (f) fclose(f); free(t); return NULL; }
    fclose(f);
    snprintf(p, sizeof p, "%s/normal_icdf_4097.f32", data_dir);
    f = fopen(p, "rb");
    if (!f || fread(t->normal_icdf, 4, 4097, f) != 4097) { if (f) fclose(f); free(t); return NULL; }
    fclose(f);
    snprintf(p, sizeof p, "%s/normal_tail_4097.f32", data_dir);
    f = fopen(p, "rb");
    if (!f || fread(t->normal_tail, 4, 4097, f) != 4097) { if (f) fclose(f); free(t); return NULL; }
    fclose(f);
    snprintf(p, sizeof p, "%s/late_thr_16.u32", data_dir);
    f = fopen(p, "rb");
    if (!f || fread(t->late_thr, 4, 16, f) != 16) { if (f) fclose(f); free(t); return NULL; }
    fclose(f);
    return t;
}
void orc_tables_free(orc_tables *t) { free(t); }
float orc_tables_cdf(const orc_tables *t, int r, int c) { return t->cdf[r][c]; }

orc_rng *orc_rng_alloc(void) { return (orc_rng *) calloc(1, sizeof(orc_rng)); }
void orc_rng_free(orc_rng *r) { free(r); }
void orc_rng_set_tick(orc_rng *r, uint32_t tick) { r->tick = tick; }

orc_station *orc_station_alloc(void) { return (orc_station *) calloc(1, sizeof(orc_station)); }
void orc_station_free(orc_station *s) { free(s); }

/* same layout as oracle/ref_driver.cpp: ref_station_scalars / ref_station_slots */
void orc_station_scalars(const orc_station *s, double *out) {
    out[0] = s->min_power;
    out[1] = s->charge_power;
    out[2] = s->max_power;
    out[3] = s->car_number;
    out[4] = s->line;
    out[5] = s->flow_in_last;
    out[6] = s->time_hole;
    out[7] = s->transformer_limit;
}
void orc_station_slots(const orc_station *s, float *out) {
    int n = s->n;
    const float *f[7] = {s->car, s->charge, s->emergency, s->power, s->soc, s->init_soc, s->target_soc};
    for (int k = 0; k < 7; k++)
        for (int i = 0; i < n; i++) out[k * n + i] = f[k][i];
    for (int i = 0; i < n; i++) {
        out[7 * n + i] = (float) s->stay_time[i];
        out[8 * n + i] = (float) s->already[i];
    }
}

orc_env *orc_env_alloc(void) { return (orc_env *) calloc(1, sizeof(orc_env)); }
void orc_env_free(orc_env *e) { free(e); }
orc_station *orc_env_station(orc_env *e, int k) { return &e->st[k]; }
orc_rng *orc_env_rng(orc_env *e) { return &e->rng; }
int orc_env_q_overflow(const orc_env *e) { return e->q_overflow; }
void orc_env_hy_table(const orc_env *e, double *out102) { memcpy(out102, e->hy_table, sizeof e->hy_table); }

/* telemetry named after the reference attributes (MGR:183-297, HYD) */
enum { ORC_TELEM_COUNT = 38 };
int orc_env_telemetry(const orc_env *e, double *out) {
    int i = 0;
    out[i++] = e->hy_act;              /* 0  self.hy_act */
    out[i++] = e->hy_flow_speed;       /* 1  hy_sys.hy_flow_speed */
    out[i++] = e->all_power_second;    /* 2  hy_sys.all_power_second */
    out[i++] = e->store_soc;           /* 3  hy_sys.sty.Store_SOC */
    out[i++] = e->capacity;            /* 4  hy_sys.sty.capacity */
    out[i++] = e->total_mass_need;     /* 5  hy_sys.hvs.total_mass_need */
    out[i++] = e->hy_use;              /* 6  hy_sys.sty.hy_use */
    out[i++] = e->not_meet;            /* 7  hy_sys.sty.not_meet */
    out[i++] = e->fc_power;            /* 8  self.fc_power */
    out[i++] = e->hy_to_use;           /* 9  hfc.hy_to_use */
    out[i++] = e->used_renew;          /* 10 self.re_used_renew */
    out[i++] = e->ev_list[0];          /* 11 self.re_ev_power_list[0] */
    out[i++] = e->ev_list[1];          /* 12 self.re_ev_power_list[1] */
    out[i++] = e->hydrogen_power_grid; /* 13 self.re_hydrogen_power */
    out[i++] = e->income;              /* 14 self.income */
    out[i++] = e->reward;              /* 15 reward */
    out[i++] = e->re_pv;               /* 16 self.re_pv_power */
    out[i++] = e->re_wd;               /* 17 self.re_wd_power */
    out[i++] = e->price_next;          /* 18 real_state[1] */
    out[i++] = (double) e->hv_arrive;  /* 19 hvs.arrive_number */
    out[i++] = (double) e->hv_line;    /* 20 hvs.line */
    out[i++] = (double) e->q_len;      /* 21 len(hvs.needed_time_list) */
    out[i++] = (double) e->pv_day;     /* 22 */
    out[i++] = (double) e->wd_day;     /* 23 */
    out[i++] = e->ev_net[0];           /* 24 ev_power_list[0] after the fuel-cell rescale (MGR:219-224) */
    out[i++] = e->ev_net[1];           /* 25 */
    out[i++] = e->ev_sum_net;          /* 26 self.real_charging_power */
    out[i++] = e->price_now;           /* 27 4 * self.re_price_dollar */
    for (int k = 0; k < 2; k++) {      /* 28-37 what make_state reads off the stations (MGR:364-368) + flow_in_number[-1] */
        out[i++] = e->st[k].min_power;
        out[i++] = e->st[k].charge_power;
        out[i++] = e->st[k].max_power;
        out[i++] = (double) e->st[k].line;
        out[i++] = (double) e->st[k].flow_in_last;
    }
    return i;
}

/* Reference constructor order with live streams (MGR:25-130): station constructors each run one
 * evs_reset (CHS:1152,1462), HySystem's 101-step sweep runs real hvs_step()s (HYD:154,168).  The
 * caller then performs the constructor's own reset() (MGR:120) with that reset's exogenous tape. */
void orc_env_init_compat_ctor(orc_env *e, const orc_config *cfg, const orc_tables *t, uint32_t glibc_seed,
                              uint32_t minstd_seed) {
    orc_env_init(e, cfg, t);
    orc_rng_seed_compat(&e->rng, glibc_seed, minstd_seed);
    orc_station_reset(&e->st[0], &e->rng, t);
    orc_station_reset(&e->st[1], &e->rng, t);
    e->capacity = cfg->init_soc * e->cap_mass;
    e->store_soc = cfg->init_soc;
    e->sys_time = 0;
    for (int i = 0; i < 101; i++) e->hy_table[i] = hy_step(e, 0.01 * i, NULL);
    e->hy_table[101] = e->hy_table[100];
    hy_reset(e);
}

/* override the action->power table (used when the reference's construction sweep was clamp-bound
 * and therefore depended on its live random FCEV demand, HYD:154,168,172-173) */
void orc_env_set_hy_table(orc_env *e, const double *in102) { memcpy(e->hy_table, in102, sizeof e->hy_table); }
