// probe_glds.hip — what does `buffer_load_dwordx4 ... lds` (LDS-DMA) write for a lane whose buffer offset is out of range?
// Build: hipcc --offload-arch=gfx950 -O2 tools/probes/probe_glds.hip -o gpurun_out/probe_glds && gpurun_out/probe_glds
// Prints, for a wave whose odd lanes are out of range: the LDS dwords of lane 0 (in range) and lane 1 (out of range) after the
// DMA into an LDS area pre-filled with 0xdeadbeef. "zeros" for lane 1 means the hardware zero-fills like it does for VGPR loads.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
__global__ void k(const float* src, unsigned* out) {
    __shared__ __attribute__((aligned(16))) unsigned lds[64 * 4 * 2];
    for (int i = threadIdx.x; i < 64 * 4 * 2; i += 64) lds[i] = 0xdeadbeefu;
    __syncthreads();
    __amdgpu_buffer_rsrc_t srd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, 0x7FFFFFFF, 0x00020000);
    int off = (threadIdx.x & 1) ? -16 : (int)threadIdx.x * 16;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(srd, (__attribute__((address_space(3))) void*)lds, 16, off, 0, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(srd, (__attribute__((address_space(3))) void*)(lds + 256), 16, off, 0, 0, 0);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * 4 * 2; i += 64) out[i] = lds[i];
}
int main() {
    std::vector<float> h(1024);
    for (int i = 0; i < 1024; ++i) h[i] = 1.0f + i;
    float* d; unsigned* o;
    hipMalloc(&d, 4096); hipMalloc(&o, 64 * 4 * 2 * 4);
    hipMemcpy(d, h.data(), 4096, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o);
    std::vector<unsigned> r(512);
    hipMemcpy(r.data(), o, 2048, hipMemcpyDeviceToHost);
    for (int l = 0; l < 4; ++l) printf("lane %d: %08x %08x %08x %08x | second: %08x\n", l, r[l * 4], r[l * 4 + 1], r[l * 4 + 2], r[l * 4 + 3], r[256 + l * 4]);
    bool zero = r[4] == 0 && r[5] == 0 && r[6] == 0 && r[7] == 0;
    bool keep = r[4] == 0xdeadbeefu;
    printf("OOB lanes: %s\n", zero ? "zeros" : (keep ? "untouched" : "other"));
    float f0; memcpy(&f0, &r[0], 4); float f2; memcpy(&f2, &r[8], 4);
    printf("lane0 first = %g (expect 1), lane2 first = %g (expect 9)\n", f0, f2);
    return 0;
}
