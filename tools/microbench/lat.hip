// microbenchmark: dependent-load latency (pointer chase) in a small table, for 1 wave and for a full grid
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
__global__ void chase(const uint32_t *tab, uint32_t *out, int steps, uint32_t mask) {
    uint32_t i = (threadIdx.x * 97u + blockIdx.x * 131u) & mask;
    for (int s = 0; s < steps; s++) i = tab[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = i;
}
__global__ void chase_u8(const uint8_t *tab, uint32_t *out, int steps, uint32_t mask) {
    uint32_t i = (threadIdx.x * 97u + blockIdx.x * 131u) & mask;
    for (int s = 0; s < steps; s++) i = (i * 5u + tab[i]) & mask;
    out[blockIdx.x * blockDim.x + threadIdx.x] = i;
}
int main() {
    for (int logn : {10, 16, 20, 24}) {
        uint32_t n = 1u << logn, mask = n - 1;
        std::vector<uint32_t> h(n);
        for (uint32_t i = 0; i < n; i++) h[i] = (i * 2654435761u + 12345u) & mask;
        uint32_t *d, *o; uint8_t *d8;
        hipMalloc(&d, n * 4); hipMalloc(&o, 1 << 24); hipMalloc(&d8, n);
        hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
        hipMemset(d8, 3, n);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int blocks : {1, 256, 2048}) {
            for (int kind = 0; kind < 2; kind++) {
                const int steps = 200;
                float ms = 0;
                for (int rep = 0; rep < 2; rep++) {
                    hipEventRecord(e0);
                    if (kind == 0) chase<<<blocks, 64>>>(d, o, steps, mask);
                    else chase_u8<<<blocks, 64>>>(d8, o, steps, mask);
                    hipEventRecord(e1); hipEventSynchronize(e1);
                    hipEventElapsedTime(&ms, e0, e1);
                }
                printf("table 2^%d x %s, %4d waves: %.1f ns per dependent load\n", logn, kind ? "u8 " : "u32", blocks, ms * 1e6 / steps);
            }
        }
        hipFree(d); hipFree(o); hipFree(d8);
    }
    return 0;
}
