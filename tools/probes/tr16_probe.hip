// Probe: exact data routing of ds_read_b64_tr_b16 on gfx950.
// LDS holds element e at index e (value = e).  Lane l supplies byte address addr[l].
// Output: 4 values per lane.  Build: hipcc --offload-arch=gfx950 -O2 tr16_probe.hip -o tr16_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef short short4v __attribute__((ext_vector_type(4)));
__global__ void k(const int* addr, short* out) {
    __shared__ __attribute__((aligned(16))) short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (short)i;
    __syncthreads();
    short4v v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)((__attribute__((address_space(3))) char*)lds + addr[threadIdx.x]));
    for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = v[j];
}
int main() {
    int* d_addr; short* d_out;
    hipMalloc(&d_addr, 64 * 4); hipMalloc(&d_out, 64 * 4 * 2);
    for (int test = 0; test < 3; ++test) {
        std::vector<int> a(64);
        for (int l = 0; l < 64; ++l) {
            if (test == 0) a[l] = l * 8;                       // canonical contiguous
            if (test == 1) a[l] = 512;                          // uniform
            if (test == 2) {                                    // 4 rows x 4 pieces, row stride 256 B, per 16-lane group
                int g = l >> 4, x = l & 15;
                a[l] = ((x >> 2) + 8 * (g >> 1)) * 256 + 2 * (16 * (g & 1) + 4 * (x & 3));
            }
        }
        hipMemcpy(d_addr, a.data(), 256, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d_addr, d_out);
        std::vector<short> o(256);
        hipMemcpy(o.data(), d_out, 512, hipMemcpyDeviceToHost);
        printf("test %d\n", test);
        for (int l = 0; l < 64; ++l) printf("  lane %2d addr %5d -> %5d %5d %5d %5d\n", l, a[l], o[l*4], o[l*4+1], o[l*4+2], o[l*4+3]);
    }
    return 0;
}
