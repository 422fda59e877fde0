// microbenchmark: cost of 32-bit integer multiplies vs adds/xors on gfx950 (wave64), and of Philox4x32-10 vs Threefry
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

template <int MODE>
__global__ void k(uint32_t *out, int iters) {
    uint32_t a = threadIdx.x * 2654435761u + blockIdx.x, b = a ^ 0x9E3779B9u, c = a + 77, d = b + 99;
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) {  // 4 independent mul_hi + 4 mul_lo per iter
            uint32_t h0 = __umulhi(0xD2511F53u, a), l0 = 0xD2511F53u * a;
            uint32_t h1 = __umulhi(0xCD9E8D57u, c), l1 = 0xCD9E8D57u * c;
            a = h1 ^ b; b = l1; c = h0 ^ d; d = l0;
            uint32_t h2 = __umulhi(0xD2511F53u, a), l2 = 0xD2511F53u * a;
            uint32_t h3 = __umulhi(0xCD9E8D57u, c), l3 = 0xCD9E8D57u * c;
            a = h3 ^ b; b = l3; c = h2 ^ d; d = l2;
        } else if (MODE == 1) {  // 8 add/xor/rot
            a += b; d ^= a; d = (d << 16) | (d >> 16);
            c += d; b ^= c; b = (b << 12) | (b >> 20);
            a += b; d ^= a; d = (d << 8) | (d >> 24);
            c += d; b ^= c; b = (b << 7) | (b >> 25);
        } else if (MODE == 2) {  // mul24
            a = __umul24(a, 0x511F53u) ^ b; b = __umul24(c, 0x9E8D57u) ^ d; c = __umul24(a, 0x2511F5u); d = __umul24(b, 0xD9E8D5u);
            a = __umul24(a, 0x511F53u) ^ b; b = __umul24(c, 0x9E8D57u) ^ d; c = __umul24(a, 0x2511F5u); d = __umul24(b, 0xD9E8D5u);
        } else {  // f64 fma x8
            double x = a, y = b;
            for (int q = 0; q < 4; q++) { x = x * 1.0000001 + y; y = y * 0.9999999 + x; }
            a = (uint32_t) x; b = (uint32_t) y;
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d;
}

int main() {
    uint32_t *d;
    hipMalloc(&d, 65536 * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 200, blocks = 8192;  // 32768 waves = 8 per SIMD... 4 per SIMD on 1024 SIMDs x2
    for (int mode = 0; mode < 4; mode++) {
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0);
            if (mode == 0) k<0><<<blocks, 256>>>(d, iters);
            if (mode == 1) k<1><<<blocks, 256>>>(d, iters);
            if (mode == 2) k<2><<<blocks, 256>>>(d, iters);
            if (mode == 3) k<3><<<blocks, 256>>>(d, iters);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep) {
                double waves = blocks * 4.0, insts = waves * iters * 8.0;
                double cyc = ms * 1e-3 * 2.2e9 * 1024;  // SIMD-cycles available
                printf("mode %d: %.3f ms, %.1f SIMD-cycles per counted op (8 ops/iter)\n", mode, ms, cyc / insts);
            }
        }
    }
    return 0;
}
