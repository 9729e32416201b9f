// Does the scalar offset of a raw buffer access take part in the hardware range check on gfx950?
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(const float* in, float* out, int soff_bytes)
{
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)in, 0, 64, 0x00020000);   // 16 floats visible
    out[threadIdx.x] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, threadIdx.x * 4, soff_bytes, 0));
}
__global__ void ks(float* buf, int soff_bytes)
{
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)buf, 0, 64, 0x00020000);
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(-1.0f), r, threadIdx.x * 4, soff_bytes, 0);
}
int main()
{
    float h[256], o[64];
    for (int i = 0; i < 256; ++i) h[i] = 100.0f + i;
    float *d_in, *d_out;
    (void)hipMalloc(&d_in, sizeof(h)); (void)hipMalloc(&d_out, sizeof(o));
    (void)hipMemcpy(d_in, h, sizeof(h), hipMemcpyHostToDevice);
    for (int soff : {0, 32, 64, 128}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d_in, d_out, soff);
        (void)hipMemcpy(o, d_out, sizeof(o), hipMemcpyDeviceToHost);
        printf("load soffset=%3d:", soff);
        for (int i = 0; i < 24; i += 1) printf(" %g", o[i]);
        printf("\n");
    }
    for (int soff : {32, 128}) {
        (void)hipMemcpy(d_in, h, sizeof(h), hipMemcpyHostToDevice);
        hipLaunchKernelGGL(ks, dim3(1), dim3(64), 0, 0, d_in, soff);
        float g[256];
        (void)hipMemcpy(g, d_in, sizeof(g), hipMemcpyDeviceToHost);
        int first = -1, last = -1;
        for (int i = 0; i < 256; ++i) if (g[i] == -1.0f) { if (first < 0) first = i; last = i; }
        printf("store soffset=%3d: elements written %d..%d\n", soff, first, last);
    }
    return 0;
}
