// TEST INFRASTRUCTURE -- not product code.
//
// Thin extern "C" driver around the UNMODIFIED reference header
//   /root/reference/evcssp_env_cpp/envs/lion_cpp20/SCP_Base/CHS.hpp
// compiled where it lies (see oracle/Makefile, target _ref/libchs_ref.so).  Nothing of the
// reference is copied: this file only #includes it and forwards calls, the way the reference's
// own Boost.Python file (lion_cpp20/main.cpp:19-290) does.  The header relies on its includer
// for the std headers below (cf. main.cpp:1-9).
//
// The reference keeps all randomness in two process-global streams (CHS.hpp:23-45): glibc rand()
// and `std::default_random_engine e`.  ref_rng_save/ref_rng_load snapshot both so that a test can
// multiplex many independent environments through this one process.
#include <fstream>
#include <string>
#include <vector>
#include <algorithm>
#include <random>
#include <cstdlib>
#include <ctime>
#include <map>
#include <cstring>
#include <sstream>

#include "SCP_Base/CHS.hpp"

namespace {
struct Mute {
    std::streambuf *old;
    std::ostringstream sink;
    Mute() : old(std::cout.rdbuf(sink.rdbuf())) {}
    ~Mute() { std::cout.rdbuf(old); }
};
char g_glibc_state[128];
bool g_state_installed = false;
void install_state(unsigned seed) {
    // TYPE_3 (128-byte) state == what rand() uses by default; initstate() makes it addressable.
    initstate(seed, g_glibc_state, sizeof g_glibc_state);
    g_state_installed = true;
}
struct Hub {
    int type;  // 0 fast, 1 slow
    FastChargeStation *f;
    SlowChargeStation *s;
    StationBase *b() { return type == 0 ? (StationBase *) f : (StationBase *) s; }
};
}  // namespace

extern "C" {

// ---- RNG control -------------------------------------------------------------------------
void ref_seed(unsigned glibc_seed, unsigned minstd_seed) {
    Change_Use_Seed(false);
    install_state(glibc_seed);  // same sequence as srand(glibc_seed)
    e.seed(minstd_seed);
}
// 128 bytes of glibc state + two ring offsets + the minstd word.
int ref_rng_state_size() { return 128 + 3 * (int) sizeof(int); }
void ref_rng_save(char *buf) {
    if (!g_state_installed) install_state(1);
    // glibc keeps fptr/rptr outside the buffer; setstate() round-trips them into word 0.
    char tmp[128];
    initstate(1, tmp, sizeof tmp);          // switch away => glibc writes rear ptr into old buffer[0]
    memcpy(buf, g_glibc_state, 128);
    setstate(g_glibc_state);                // switch back
    std::ostringstream os;
    os << e;
    unsigned v = (unsigned) std::stoul(os.str());
    memcpy(buf + 128, &v, sizeof v);
}
void ref_rng_load(const char *buf) {
    char tmp[128];
    initstate(1, tmp, sizeof tmp);
    memcpy(g_glibc_state, buf, 128);
    setstate(g_glibc_state);
    g_state_installed = true;
    unsigned v;
    memcpy(&v, buf + 128, sizeof v);
    e.seed(v);
}
void ref_change_use_seed(int v) { Change_Use_Seed(v != 0); }   // binding of CHS.hpp:27 (main.cpp:21)
int ref_c_rand() { return rand(); }
unsigned ref_minstd_next() { return (unsigned) e(); }
float ref_uniform_rand(float a, float b) { return RandomUtil::uniform_rand(a, b); }
float ref_mk_soc() { return CarArriveRandom::mk_soc(); }
int ref_mk_late_time(int fast) { return CarArriveRandom::mk_late_time(fast ? "fast" : "slow"); }
int ref_init_station_car_number(int mu) { return CarArriveRandom::init_station_car_number(mu, 3); }

// ---- arrival table -------------------------------------------------------------------------
int ref_cdf_rows() { return (int) car_flow_possibility_list.size(); }
int ref_cdf_cols(int r) { return (int) car_flow_possibility_list[r].size(); }
double ref_cdf(int r, int c) { return car_flow_possibility_list[r][c]; }
int ref_give_car_number(int t) { return PoissonNumber::give_car_number_wrt_poisson(t); }
int ref_ev_fast(int t) { return PoissonNumber::ev_car_number_wrt_poisson_fast(t); }
int ref_ev_slow(int t) { return PoissonNumber::ev_car_number_wrt_poisson_slow(t); }
int ref_hv(int t, float pin, float perm) { return PoissonNumber::hv_car_number_wrt_poisson(t, pin, perm); }
float ref_parse_float(const char *s) { return Read2Vector::string_to_float(std::string(s)); }

// ---- charge curves: which = 0 time_to_power, 1 time_to_soc, 2 soc_to_time -------------------
float ref_curve_slow(int which, float x, int constant_power) {
    bool c = constant_power != 0;
    if (which == 0) return UtilSlow::slow_time_to_power(x, c);
    if (which == 1) return UtilSlow::slow_time_to_soc(x, c);
    return UtilSlow::slow_soc_to_time(x, c);
}
float ref_curve_fast(int which, float x, int constant_power) {
    bool c = constant_power != 0;
    if (which == 0) return UtilFast::fast_time_to_power(x, c);
    if (which == 1) return UtilFast::fast_time_to_soc(x, c);
    return UtilFast::fast_soc_to_time(x, c);
}

// ---- stations ------------------------------------------------------------------------------
void *ref_station_new(int type, int piles, int wait, int constant_charging) {
    Mute m;
    Hub *h = new Hub{type, nullptr, nullptr};
    if (type == 0) h->f = new FastChargeStation(piles, wait != 0, constant_charging != 0);
    else h->s = new SlowChargeStation(piles, wait != 0, constant_charging != 0);
    return h;
}
void ref_station_free(void *p) {
    Hub *h = (Hub *) p;
    delete h->f;
    delete h->s;
    delete h;
}
void ref_station_reset(void *p) {
    Mute m;
    Hub *h = (Hub *) p;
    if (h->type == 0) h->f->evs_reset(); else h->s->evs_reset();
}
void ref_station_step(void *p, const float *actions, int n) {
    Mute m;
    Hub *h = (Hub *) p;
    std::vector<float> v(actions, actions + n);
    if (h->type == 0) h->f->evs_step_wrapper3(v); else h->s->evs_step_wrapper3(v);
}
// scalar-load control mode (CHS.hpp:1169-1186 / 1480-1497)
void ref_station_step_load(void *p, float load) {
    Mute m;
    Hub *h = (Hub *) p;
    if (h->type == 0) h->f->evs_step_wrapper1(load); else h->s->evs_step_wrapper1(load);
}
// out[0..5] = min_power, charge_power, max_power, car_number, line, flow_in_number.back(),
// out[6] = station_time_hole, out[7] = transformer_limit
void ref_station_scalars(void *p, double *out) {
    Hub *h = (Hub *) p;
    StationBase *b = h->b();
    out[0] = b->min_power;
    out[1] = b->charge_power;
    out[2] = b->max_power;
    out[3] = b->car_number;
    out[4] = b->line;
    out[5] = b->flow_in_number.empty() ? 0 : b->flow_in_number.back();
    out[6] = b->station_time_hole;
    out[7] = h->type == 0 ? h->f->transformer_limit : h->s->transformer_limit;
}
// per slot, field-major: car, charge, emergency, power, soc, init_soc, target_soc (situation map,
// CHS.hpp:204-231) then stay_time, already_stay_time (pile fields, CHS.hpp:245-246)
void ref_station_slots(void *p, float *out /* [9][piles] */) {
    Hub *h = (Hub *) p;
    StationBase *b = h->b();
    int n = b->charge_number;
    static const char *names[7] = {"car", "charge", "emergency", "power", "soc", "init_soc", "target_soc"};
    for (int f = 0; f < 7; f++)
        for (int i = 0; i < n; i++) out[f * n + i] = b->situation[names[f]][i];
    for (int i = 0; i < n; i++) {
        ChargePositionBase *c;
        if (h->type == 0) c = &h->f->positions.at("FP" + std::to_string(i));
        else c = &h->s->positions.at("SP" + std::to_string(i));
        out[7 * n + i] = (float) c->stay_time;
        out[8 * n + i] = (float) c->already_stay_time;
    }
}

}  // extern "C"
