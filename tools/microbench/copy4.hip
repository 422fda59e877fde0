// calibration kernel for the rocprofv3 FETCH_SIZE / WRITE_SIZE counters with THIS path's access shape:
// one dword (and one byte) per lane, contiguous per wave.  Known bytes: reads n*4 + n, writes n*4 + n.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void copy4(const float *a, const uint8_t *b, float *c, uint8_t *d, size_t n) {
    size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { c[i] = a[i] + 1.0f; d[i] = (uint8_t) (b[i] + 1); }
}
int main() {
    const size_t n = (size_t) 96 << 20;  // 96 Mi elements: 384 MiB + 96 MiB each way, beyond the 256 MiB Infinity Cache
    float *a, *c; uint8_t *b, *d;
    hipMalloc(&a, n * 4); hipMalloc(&c, n * 4); hipMalloc(&b, n); hipMalloc(&d, n);
    hipMemset(a, 0, n * 4); hipMemset(b, 0, n);
    for (int r = 0; r < 3; r++) copy4<<<(unsigned) ((n + 255) / 256), 256>>>(a, b, c, d, n);
    hipDeviceSynchronize();
    printf("copy4: read bytes %zu, written bytes %zu per launch\n", n * 5, n * 5);
    return 0;
}
