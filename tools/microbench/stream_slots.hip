// The slot kernel's memory streams without its work: per slot 8 B of state in, 4 B of action in (rows of S + 2 floats), 8 B of state
// out; per (env, station) 4 B in and 16 B out.  65 536 envs x 45 slots, launched back to back like the step (the 47 MB of state
// stay in the Infinity Cache from launch to launch, the action batches cycle through 8 x 12 MB).  What it prints is the floor a
// kernel with these streams and nothing else reaches on this chip, for four shapes of workgroup -- and then the same with the
// step's one scattered read added: 16 bytes of a 768 KB table (2 x 2048 class rows of 256 B, resident in L2) for 55 % of the slots.
//   hipcc --offload-arch=gfx950 -O3 -o stream_slots tools/microbench/stream_slots.hip && ./stream_slots
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t mix(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

template <int BLOCK, int T, int GATHER>
__global__ __launch_bounds__(BLOCK) void k_stream(u32x2 *__restrict__ state, const float *__restrict__ act, const uint32_t *__restrict__ pk,
                                                  u32x4 *__restrict__ rec, int n_envs, int S, int epb, uint32_t magic,
                                                  const u32x4 *__restrict__ table, uint32_t salt) {
    const int tid = threadIdx.x, env_first = blockIdx.x * epb;
    u32x2 s[T];
    float a[T];
    bool ok[T];
    uint32_t idx[T];
#pragma unroll
    for (int j = 0; j < T; j++) {
        const int v = tid + j * BLOCK, e = (int) (((uint32_t) v * magic) >> 20), env = env_first + e;
        ok[j] = e < epb && env < n_envs;
        idx[j] = (uint32_t) env_first * (uint32_t) S + (uint32_t) v;
        if (ok[j]) {
            s[j] = state[idx[j]];
            if (GATHER == 5) {  // the action row reduced to one bit per slot: 8 bytes per env
                const uint64_t w = ((const uint64_t *) act)[env];
                a[j] = ((w >> ((uint32_t) v - (uint32_t) e * (uint32_t) S)) & 1ull) ? 1.0f : -1.0f;
            } else if (GATHER == 6) a[j] = __builtin_nontemporal_load(act + idx[j] + 2u * (uint32_t) env);  // the row is read once, ever
            else a[j] = act[idx[j] + 2u * (uint32_t) env];
        }
    }
    if (GATHER) {  // the occupied slots' class-row read: address known only when the state is here
        u32x4 r[T];
#pragma unroll
        for (int j = 0; j < T; j++) {
            const uint32_t h = mix(idx[j] ^ salt ^ s[j].y);
            r[j] = u32x4{0u, 0u, 0u, 0u};
            if (ok[j] && h % 100u < 55u) {
                const char *p = (const char *) table + ((h >> 8) % 4096u) * 256u + ((h >> 24) % 27u) * 8u;
                if (GATHER == 1 || GATHER == 5 || GATHER == 6) r[j] = *(const u32x4 *) p;                                  // 16 bytes, as the step reads them
                else if (GATHER == 2) { const u32x2 q = *(const u32x2 *) p; r[j].x = q.x; r[j].w = q.y; }  // 8 bytes
                else if (GATHER == 3) r[j] = __builtin_nontemporal_load((const u32x4 *) p);  // 16 bytes, nt
                else {                                                                        // 16 bytes past the vector L1 (agent scope)
                    const uint64_t q0 = __hip_atomic_load((const uint64_t *) p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const uint64_t q1 = __hip_atomic_load((const uint64_t *) p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    r[j].x = (uint32_t) q0; r[j].w = (uint32_t) (q1 >> 32);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < T; j++) s[j].y += r[j].x + r[j].w;
    }
#pragma unroll
    for (int j = 0; j < T; j++)
        if (ok[j]) {
            s[j].x += a[j] > 0.0f ? 1u : 0u;
            state[idx[j]] = s[j];
        }
    if (tid < 2 * epb) {
        const int env = env_first + (tid >> 1);
        if (env < n_envs) {
            const uint32_t u = (uint32_t) (tid & 1) * (uint32_t) n_envs + (uint32_t) env;
            const uint32_t p = pk[u];
            rec[u] = u32x4{p, p + 1u, p + 2u, p + 3u};
        }
    }
}

// What a 4-byte slot state would buy (and cost): 4 B in / out per slot instead of 8, the class-row gather as before, plus a 4-byte
// read of a 4 KB table (the target time behind a 10-bit level: resident in the vector L1) for the same slots.
template <int BLOCK, int T, bool BITS>
__global__ __launch_bounds__(BLOCK) void k_stream4(uint32_t *__restrict__ state, const float *__restrict__ act, const uint32_t *__restrict__ pk,
                                                   u32x4 *__restrict__ rec, int n_envs, int S, int epb, uint32_t magic,
                                                   const u32x4 *__restrict__ table, const float *__restrict__ ttab, uint32_t salt) {
    const int tid = threadIdx.x, env_first = blockIdx.x * epb;
    uint32_t s[T], idx[T];
    float a[T];
    bool ok[T];
#pragma unroll
    for (int j = 0; j < T; j++) {
        const int v = tid + j * BLOCK, e = (int) (((uint32_t) v * magic) >> 20), env = env_first + e;
        ok[j] = e < epb && env < n_envs;
        idx[j] = (uint32_t) env_first * (uint32_t) S + (uint32_t) v;
        if (ok[j]) {
            s[j] = state[idx[j]];
            if (BITS) {
                const uint64_t w = ((const uint64_t *) act)[env];
                a[j] = ((w >> ((uint32_t) v - (uint32_t) e * (uint32_t) S)) & 1ull) ? 1.0f : -1.0f;
            } else a[j] = act[idx[j] + 2u * (uint32_t) env];
        }
    }
    u32x4 r[T];
    float tt[T];
#pragma unroll
    for (int j = 0; j < T; j++) {
        const uint32_t h = mix(idx[j] ^ salt ^ s[j]);
        r[j] = u32x4{0u, 0u, 0u, 0u};
        tt[j] = 0.0f;
        if (ok[j] && h % 100u < 55u) {
            r[j] = *(const u32x4 *) ((const char *) table + ((h >> 8) % 4096u) * 256u + ((h >> 24) % 27u) * 8u);
            tt[j] = ttab[(h >> 4) % 1000u];
        }
    }
#pragma unroll
    for (int j = 0; j < T; j++)
        if (ok[j]) state[idx[j]] = s[j] + r[j].x + r[j].w + (tt[j] + a[j] > 0.0f ? 1u : 0u);
    if (tid < 2 * epb) {
        const int env = env_first + (tid >> 1);
        if (env < n_envs) {
            const uint32_t u = (uint32_t) (tid & 1) * (uint32_t) n_envs + (uint32_t) env;
            const uint32_t p = pk[u];
            rec[u] = u32x4{p, p + 1u, p + 2u, p + 3u};
        }
    }
}

// The 8 bytes of a slot as TWO arrays of 4: both read, only the first written back (the target time of a car never changes while it
// stays: only its admission writes it) -- 12 bytes of state traffic per slot instead of 16, no change of format.
template <int BLOCK, int T>
__global__ __launch_bounds__(BLOCK) void k_stream_split(uint32_t *__restrict__ w0a, const uint32_t *__restrict__ w1a, const float *__restrict__ act,
                                                       const uint32_t *__restrict__ pk, u32x4 *__restrict__ rec, int n_envs, int S, int epb,
                                                       uint32_t magic, const u32x4 *__restrict__ table, uint32_t salt) {
    const int tid = threadIdx.x, env_first = blockIdx.x * epb;
    uint32_t w0[T], w1[T], idx[T];
    float a[T];
    bool ok[T];
#pragma unroll
    for (int j = 0; j < T; j++) {
        const int v = tid + j * BLOCK, e = (int) (((uint32_t) v * magic) >> 20), env = env_first + e;
        ok[j] = e < epb && env < n_envs;
        idx[j] = (uint32_t) env_first * (uint32_t) S + (uint32_t) v;
        if (ok[j]) {
            w0[j] = w0a[idx[j]];
            w1[j] = w1a[idx[j]];
            a[j] = act[idx[j] + 2u * (uint32_t) env];
        }
    }
    u32x4 r[T];
#pragma unroll
    for (int j = 0; j < T; j++) {
        const uint32_t h = mix(idx[j] ^ salt ^ w0[j]);
        r[j] = u32x4{0u, 0u, 0u, 0u};
        if (ok[j] && h % 100u < 55u) r[j] = *(const u32x4 *) ((const char *) table + ((h >> 8) % 4096u) * 256u + ((h >> 24) % 27u) * 8u);
    }
#pragma unroll
    for (int j = 0; j < T; j++)
        if (ok[j]) w0a[idx[j]] = w0[j] + r[j].x + r[j].w + w1[j] + (a[j] > 0.0f ? 1u : 0u);
    if (tid < 2 * epb) {
        const int env = env_first + (tid >> 1);
        if (env < n_envs) {
            const uint32_t u = (uint32_t) (tid & 1) * (uint32_t) n_envs + (uint32_t) env;
            const uint32_t p = pk[u];
            rec[u] = u32x4{p, p + 1u, p + 2u, p + 3u};
        }
    }
}

template <int BLOCK, int T>
static void run_split(const char *name, u32x2 *state, float **acts, uint32_t *pk, u32x4 *rec, int N, int S, const u32x4 *table) {
    const int epb = BLOCK * T / S;
    const uint32_t magic = (1u << 20) / (uint32_t) S + 1u;
    const int nb = (N + epb - 1) / epb;
    uint32_t *w0a = (uint32_t *) state, *w1a = (uint32_t *) state + (size_t) N * S;
    hipStream_t st;
    hipStreamCreate(&st);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int i = 0; i < 50; i++) hipLaunchKernelGGL((k_stream_split<BLOCK, T>), dim3(nb), dim3(BLOCK), 0, st, w0a, w1a, acts[i & 7], pk, rec, N, S, epb, magic, table, (uint32_t) i);
    const int R = 2000;
    hipEventRecord(e0, st);
    for (int i = 0; i < R; i++) hipLaunchKernelGGL((k_stream_split<BLOCK, T>), dim3(nb), dim3(BLOCK), 0, st, w0a, w1a, acts[i & 7], pk, rec, N, S, epb, magic, table, (uint32_t) i * 2654435761u);
    hipEventRecord(e1, st);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-20s the slot's two words as two arrays, both read, one written + gather 16 %5d workgroups: %.2f us per launch\n", name, nb, ms / R * 1e3);
    hipStreamDestroy(st);
}

// Two ADJACENT slots per lane: one 16-byte state load and store and one 8-byte action load per lane instead of two of each
// (the action pair is read as if no env boundary fell between the two slots: a floor, not a layout proposal).
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_stream_adj(u32x4 *__restrict__ state2, const float *__restrict__ act, const uint32_t *__restrict__ pk,
                                                     u32x4 *__restrict__ rec, int n_envs, int S, int epb, uint32_t magic,
                                                     const u32x4 *__restrict__ table, uint32_t salt) {
    const int tid = threadIdx.x, env_first = blockIdx.x * epb;
    const int v = 2 * tid, e = (int) (((uint32_t) v * magic) >> 20), env = env_first + e;
    const bool ok = e < epb && env < n_envs;
    const uint32_t idx = (uint32_t) env_first * (uint32_t) S + (uint32_t) v;
    u32x4 s = {0u, 0u, 0u, 0u};
    float a0 = 0.0f, a1 = 0.0f;
    if (ok) {
        s = *(const u32x4 *) ((const char *) state2 + (size_t) idx * 8u);
        const float *ap = act + idx + 2u * (uint32_t) env;
        a0 = ap[0];
        a1 = ap[1];
    }
    u32x4 r0 = {0u, 0u, 0u, 0u}, r1 = {0u, 0u, 0u, 0u};
    const uint32_t h0 = mix(idx ^ salt ^ s.y), h1 = mix((idx + 1u) ^ salt ^ s.w);
    if (ok && h0 % 100u < 55u) r0 = *(const u32x4 *) ((const char *) table + ((h0 >> 8) % 4096u) * 256u + ((h0 >> 24) % 27u) * 8u);
    if (ok && h1 % 100u < 55u) r1 = *(const u32x4 *) ((const char *) table + ((h1 >> 8) % 4096u) * 256u + ((h1 >> 24) % 27u) * 8u);
    if (ok) {
        s.x += a0 > 0.0f ? 1u : 0u;
        s.y += r0.x + r0.w;
        s.z += a1 > 0.0f ? 1u : 0u;
        s.w += r1.x + r1.w;
        *(u32x4 *) ((char *) state2 + (size_t) idx * 8u) = s;
    }
    if (tid < 2 * epb) {
        const int env2 = env_first + (tid >> 1);
        if (env2 < n_envs) {
            const uint32_t u = (uint32_t) (tid & 1) * (uint32_t) n_envs + (uint32_t) env2;
            const uint32_t p = pk[u];
            rec[u] = u32x4{p, p + 1u, p + 2u, p + 3u};
        }
    }
}

template <int BLOCK>
static void run_adj(const char *name, u32x2 *state, float **acts, uint32_t *pk, u32x4 *rec, int N, int S, const u32x4 *table) {
    const int epb = BLOCK * 2 / S;
    const uint32_t magic = (1u << 20) / (uint32_t) S + 1u;
    const int nb = (N + epb - 1) / epb;
    hipStream_t st;
    hipStreamCreate(&st);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int i = 0; i < 50; i++) hipLaunchKernelGGL((k_stream_adj<BLOCK>), dim3(nb), dim3(BLOCK), 0, st, (u32x4 *) state, acts[i & 7], pk, rec, N, S, epb, magic, table, (uint32_t) i);
    const int R = 2000;
    hipEventRecord(e0, st);
    for (int i = 0; i < R; i++) hipLaunchKernelGGL((k_stream_adj<BLOCK>), dim3(nb), dim3(BLOCK), 0, st, (u32x4 *) state, acts[i & 7], pk, rec, N, S, epb, magic, table, (uint32_t) i * 2654435761u);
    hipEventRecord(e1, st);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-20s two ADJACENT slots per lane (16-byte state accesses) + gather 16 %5d workgroups: %.2f us per launch\n", name, nb, ms / R * 1e3);
    hipStreamDestroy(st);
}

// FOUR ADJACENT 4-byte slots per lane: one 16-byte state load and store per lane instead of four 4-byte ones, the four actions as one
// (unaligned) 16-byte load as if no env boundary fell between them -- a floor for a slot kernel whose lanes own runs of consecutive
// slots instead of strided ones, not a layout proposal.  Same scattered reads per slot as k_stream4.
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_stream4_adj(u32x4 *__restrict__ state4, const float *__restrict__ act, const uint32_t *__restrict__ pk,
                                                      u32x4 *__restrict__ rec, int n_envs, int S, int epb, uint32_t magic,
                                                      const u32x4 *__restrict__ table, const float *__restrict__ ttab, uint32_t salt) {
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const int tid = threadIdx.x, env_first = blockIdx.x * epb;
    const int v = 4 * tid, e = (int) (((uint32_t) v * magic) >> 20), env = env_first + e;
    const bool ok = e < epb && env < n_envs;
    const uint32_t idx = (uint32_t) env_first * (uint32_t) S + (uint32_t) v;
    u32x4 s = {0u, 0u, 0u, 0u};
    f32x4 a = {0.0f, 0.0f, 0.0f, 0.0f};
    if (ok) {
        s = *(const u32x4 *) ((const char *) state4 + (size_t) idx * 4u);
        a = *(const f32x4 *) (act + idx + 2u * (uint32_t) env);
    }
    uint32_t sw[4] = {s.x, s.y, s.z, s.w};
    float aw[4] = {a.x, a.y, a.z, a.w};
    u32x4 r[4];
    float tt[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const uint32_t h = mix((idx + (uint32_t) j) ^ salt ^ sw[j]);
        r[j] = u32x4{0u, 0u, 0u, 0u};
        tt[j] = 0.0f;
        if (ok && h % 100u < 55u) {
            r[j] = *(const u32x4 *) ((const char *) table + ((h >> 8) % 4096u) * 256u + ((h >> 24) % 27u) * 8u);
            tt[j] = ttab[(h >> 4) % 1000u];
        }
    }
    if (ok) {
#pragma unroll
        for (int j = 0; j < 4; j++) sw[j] += r[j].x + r[j].w + (tt[j] + aw[j] > 0.0f ? 1u : 0u);
        *(u32x4 *) ((char *) state4 + (size_t) idx * 4u) = u32x4{sw[0], sw[1], sw[2], sw[3]};
    }
    if (tid < 2 * epb) {
        const int env2 = env_first + (tid >> 1);
        if (env2 < n_envs) {
            const uint32_t u = (uint32_t) (tid & 1) * (uint32_t) n_envs + (uint32_t) env2;
            const uint32_t p = pk[u];
            rec[u] = u32x4{p, p + 1u, p + 2u, p + 3u};
        }
    }
}

template <int BLOCK>
static void run4_adj(const char *name, uint32_t *state, float **acts, uint32_t *pk, u32x4 *rec, int N, int S, const u32x4 *table, const float *ttab) {
    const int epb = BLOCK * 4 / S;
    const uint32_t magic = (1u << 20) / (uint32_t) S + 1u;
    const int nb = (N + epb - 1) / epb;
    hipStream_t st;
    hipStreamCreate(&st);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int i = 0; i < 50; i++) hipLaunchKernelGGL((k_stream4_adj<BLOCK>), dim3(nb), dim3(BLOCK), 0, st, (u32x4 *) state, acts[i & 7], pk, rec, N, S, epb, magic, table, ttab, (uint32_t) i);
    const int R = 2000;
    hipEventRecord(e0, st);
    for (int i = 0; i < R; i++) hipLaunchKernelGGL((k_stream4_adj<BLOCK>), dim3(nb), dim3(BLOCK), 0, st, (u32x4 *) state, acts[i & 7], pk, rec, N, S, epb, magic, table, ttab, (uint32_t) i * 2654435761u);
    hipEventRecord(e1, st);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-20s 4-byte state, FOUR ADJACENT slots per lane (16-byte accesses) + gathers %5d workgroups: %.2f us per launch\n", name, nb, ms / R * 1e3);
    hipStreamDestroy(st);
}

template <int BLOCK, int T, bool BITS>
static void run4(const char *name, uint32_t *state, float **acts, uint32_t *pk, u32x4 *rec, int N, int S, const u32x4 *table, const float *ttab) {
    const int epb = BLOCK * T / S;
    const uint32_t magic = (1u << 20) / (uint32_t) S + 1u;
    const int nb = (N + epb - 1) / epb;
    hipStream_t st;
    hipStreamCreate(&st);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int i = 0; i < 50; i++) hipLaunchKernelGGL((k_stream4<BLOCK, T, BITS>), dim3(nb), dim3(BLOCK), 0, st, state, acts[i & 7], pk, rec, N, S, epb, magic, table, ttab, (uint32_t) i);
    const int R = 2000;
    hipEventRecord(e0, st);
    for (int i = 0; i < R; i++) hipLaunchKernelGGL((k_stream4<BLOCK, T, BITS>), dim3(nb), dim3(BLOCK), 0, st, state, acts[i & 7], pk, rec, N, S, epb, magic, table, ttab, (uint32_t) i * 2654435761u);
    hipEventRecord(e1, st);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-20s 4-byte state + gather 16 + 4 B of a 4 KB table%s %5d workgroups: %.2f us per launch\n", name, BITS ? ", bit actions" : "", nb, ms / R * 1e3);
    hipStreamDestroy(st);
}

template <int BLOCK, int T, int GATHER>
static void run(const char *name, u32x2 *state, float **acts, uint32_t *pk, u32x4 *rec, int N, int S, const u32x4 *table) {
    const int epb = BLOCK * T / S;
    const uint32_t magic = (1u << 20) / (uint32_t) S + 1u;
    const int nb = (N + epb - 1) / epb;
    hipStream_t st;
    hipStreamCreate(&st);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int i = 0; i < 50; i++) hipLaunchKernelGGL((k_stream<BLOCK, T, GATHER>), dim3(nb), dim3(BLOCK), 0, st, state, acts[i & 7], pk, rec, N, S, epb, magic, table, (uint32_t) i * 2654435761u);
    const int R = 2000;
    hipEventRecord(e0, st);
    for (int i = 0; i < R; i++) hipLaunchKernelGGL((k_stream<BLOCK, T, GATHER>), dim3(nb), dim3(BLOCK), 0, st, state, acts[i & 7], pk, rec, N, S, epb, magic, table, (uint32_t) i * 2654435761u);
    hipEventRecord(e1, st);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double) N * (S * 20.0 + 2 * 20.0);
    printf("%-20s %s %5d workgroups: %.2f us per launch (back to back), %.0f GB/s of its %.1f MB of streams\n", name,
           GATHER == 0 ? "            " : GATHER == 1 ? "+ gather 16 " : GATHER == 2 ? "+ gather 8  " : GATHER == 3 ? "+ gather nt " : GATHER == 4 ? "+ gather sc1" : GATHER == 5 ? "+ gather 16, bit actions" : "+ gather 16, nt action loads", nb, ms / R * 1e3, bytes / (ms / R * 1e-3) / 1e9, bytes / 1e6);
    hipStreamDestroy(st);
}

int main(int argc, char **argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 65536, S = argc > 2 ? atoi(argv[2]) : 45;
    u32x2 *state;
    uint32_t *pk;
    u32x4 *rec;
    float *acts[8];
    hipMalloc(&state, (size_t) N * S * 8);
    hipMemset(state, 0, (size_t) N * S * 8);
    hipMalloc(&pk, (size_t) N * 2 * 4);
    hipMemset(pk, 0, (size_t) N * 2 * 4);
    hipMalloc(&rec, (size_t) N * 2 * 16);
    for (float *&a : acts) {
        hipMalloc(&a, (size_t) N * (S + 2) * 4);
        hipMemset(a, 0, (size_t) N * (S + 2) * 4);
    }
    hipDeviceSynchronize();
    printf("%d envs x %d slots: state %.1f MB, one action batch %.1f MB\n", N, S, N * (double) S * 8 / 1e6, N * (S + 2.0) * 4 / 1e6);
    u32x4 *table;
    hipMalloc(&table, 4096 * 256 + 256);
    hipMemset(table, 1, 4096 * 256 + 256);
    hipDeviceSynchronize();
    run<256, 1, 0>("256 lanes x 1 slot", state, acts, pk, rec, N, S, table);
    run<256, 2, 0>("256 lanes x 2 slots", state, acts, pk, rec, N, S, table);
    run<256, 4, 0>("256 lanes x 4 slots", state, acts, pk, rec, N, S, table);
    run<512, 2, 0>("512 lanes x 2 slots", state, acts, pk, rec, N, S, table);
    run<256, 1, 1>("256 lanes x 1 slot", state, acts, pk, rec, N, S, table);
    run<256, 2, 1>("256 lanes x 2 slots", state, acts, pk, rec, N, S, table);
    run<256, 4, 1>("256 lanes x 4 slots", state, acts, pk, rec, N, S, table);
    run<512, 2, 1>("512 lanes x 2 slots", state, acts, pk, rec, N, S, table);
    run<256, 2, 2>("256 lanes x 2 slots", state, acts, pk, rec, N, S, table);
    run<256, 2, 3>("256 lanes x 2 slots", state, acts, pk, rec, N, S, table);
    run<256, 2, 4>("256 lanes x 2 slots", state, acts, pk, rec, N, S, table);
    run<256, 2, 5>("256 lanes x 2 slots", state, acts, pk, rec, N, S, table);
    run<256, 2, 6>("256 lanes x 2 slots", state, acts, pk, rec, N, S, table);
    run<256, 2, 1>("256 lanes x 2 slots", state, acts, pk, rec, N, S, table);
    float *ttab;
    hipMalloc(&ttab, 4096);
    hipMemset(ttab, 0, 4096);
    hipDeviceSynchronize();
    run_adj<256>("256 lanes x 2 slots", state, acts, pk, rec, N, S, table);
    run_split<256, 2>("256 lanes x 2 slots", state, acts, pk, rec, N, S, table);
    run4<256, 2, false>("256 lanes x 2 slots", (uint32_t *) state, acts, pk, rec, N, S, table, ttab);
    run4<256, 2, true>("256 lanes x 2 slots", (uint32_t *) state, acts, pk, rec, N, S, table, ttab);
    run4<256, 4, false>("256 lanes x 4 slots", (uint32_t *) state, acts, pk, rec, N, S, table, ttab);
    run4<512, 4, false>("512 lanes x 4 slots", (uint32_t *) state, acts, pk, rec, N, S, table, ttab);
    run4_adj<256>("256 lanes x 4 adj", (uint32_t *) state, acts, pk, rec, N, S, table, ttab);
    run4_adj<512>("512 lanes x 4 adj", (uint32_t *) state, acts, pk, rec, N, S, table, ttab);
    run4<256, 2, false>("256 lanes x 2 slots", (uint32_t *) state, acts, pk, rec, N, S, table, ttab);
    return 0;
}
