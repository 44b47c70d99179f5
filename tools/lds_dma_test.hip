// Probe of buffer_load ... lds (LDS-DMA) semantics on gfx950 used by the conv staging design:
//  (1) out-of-range lanes write 0 to LDS?  (2) LDS destinations above 64 KiB work?  (3) dwordx4 form.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef __attribute__((address_space(3))) void* lds_ptr_t;

__global__ void k(const float* p, int n, float* out) {
    extern __shared__ __attribute__((aligned(16))) float s[];   // 96 KiB
    const int tid = threadIdx.x, wave = tid >> 6;
    for (int i = tid; i < 24576; i += 256) s[i] = -1.f;
    __syncthreads();
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, n * 4, 0x00020000);
    // region A at float 0: 256 dwords, lanes >= 200 out of range (n = 200)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr_t)(s + wave * 64), 4, tid * 4, 0, 0, 0);
    // region B at float 20000 (80 KB offset): same
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr_t)(s + 20000 + wave * 64), 4, tid * 4, 0, 0, 0);
    // region C at float 4096: dwordx4, lanes cover 1024 floats; only first 200 in range
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr_t)(s + 4096 + wave * 256), 16, tid * 16, 0, 0, 0);
    // region D: half the lanes inactive
    if (tid & 1) __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr_t)(s + 8192 + wave * 64), 4, tid * 4, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = tid; i < 256; i += 256) {
        out[i] = s[i];
        out[256 + i] = s[20000 + i];
        out[1536 + i] = s[8192 + i];
    }
    for (int i = tid; i < 1024; i += 256) out[512 + i] = s[4096 + i];
}

int main() {
    const int n = 200;
    std::vector<float> h(1024);
    for (int i = 0; i < 1024; ++i) h[i] = 1000.f + i;
    float *d, *o;
    hipMalloc(&d, 4096);
    hipMalloc(&o, 2048 * 4);
    hipMemcpy(d, h.data(), 4096, hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 98304);
    k<<<1, 256, 98304>>>(d, n, o);
    std::vector<float> r(2048);
    hipMemcpy(r.data(), o, 2048 * 4, hipMemcpyDeviceToHost);
    printf("err %s\n", hipGetErrorString(hipGetLastError()));
    printf("A: [0]=%g [199]=%g [200]=%g [255]=%g\n", r[0], r[199], r[200], r[255]);
    printf("B(80KB): [0]=%g [199]=%g [200]=%g\n", r[256], r[256 + 199], r[256 + 200]);
    printf("C(x4): [0]=%g [3]=%g [4]=%g [199]=%g [200]=%g [203]=%g [1023]=%g\n", r[512], r[515], r[516], r[512 + 199], r[512 + 200], r[512 + 203], r[512 + 1023]);
    printf("D(half lanes): [0]=%g [1]=%g [2]=%g [3]=%g\n", r[1536], r[1537], r[1538], r[1539]);
    return 0;
}
