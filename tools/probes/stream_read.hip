// Probe: plain contiguous read of a large buffer (grid-stride, 16 B per lane, 4 loads in flight per thread), the
// simplest possible HBM reader: what this MI355X delivers to ANY kernel, to put the gather numbers in context.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int uintx4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void rd(const uintx4* p, size_t n, unsigned* out) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, stride = (size_t)gridDim.x * 256;
    unsigned acc = 0;
    for (; i + 3 * stride < n; i += 4 * stride) {
        uintx4 a = p[i], b = p[i + stride], c = p[i + 2 * stride], d = p[i + 3 * stride];
        acc += a.x ^ b.y ^ c.z ^ d.w;
    }
    for (; i < n; i += stride) acc += p[i].x;
    if (acc == 0x12345678u) out[0] = acc;
}
int main() {
    const size_t bytes = (size_t)4 << 30;
    char* d; unsigned* o;
    hipMalloc(&d, bytes); hipMalloc(&o, 4); hipMemset(d, 1, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wgs : {256, 512, 1024, 2048, 4096}) {
        for (size_t sz : {(size_t)172 << 20, (size_t)1 << 30, bytes}) {
            float best = 1e9;
            for (int rep = 0; rep < 4; ++rep) {
                // rotate the start so that a 172 MB read does not come from the 256 MB Infinity Cache
                const size_t off = sz < bytes ? ((size_t)rep * (sz + ((size_t)300 << 20))) % (bytes - sz) / 16 * 16 : 0;
                hipEventRecord(e0);
                hipLaunchKernelGGL(rd, dim3(wgs), dim3(256), 0, 0, (const uintx4*)(d + off), sz / 16, o);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (rep && ms < best) best = ms;
            }
            printf("%5d WGs, %4zu MB: %.1f us  %.2f TB/s\n", wgs, sz >> 20, best * 1e3, sz / (best * 1e-3) / 1e12);
        }
    }
    return 0;
}
