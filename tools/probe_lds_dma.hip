// Hardware probe (not part of the product): what does an LDS-DMA buffer load write for an out-of-range lane?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
typedef __attribute__((address_space(3))) void lds_void;
__global__ void probe(const unsigned* src, unsigned nbytes, unsigned* out) {
    __shared__ __attribute__((aligned(16))) unsigned lds[64 * 4];
    const int l = threadIdx.x;
    for (int i = 0; i < 4; ++i) lds[l * 4 + i] = 0xABABABABu;
    __syncthreads();
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, nbytes, 0x00020000);
    // lanes 0..31 in range, lanes 32..63 out of range (offset beyond num_records)
    const unsigned voff = l < 32 ? l * 16 : 0x40000000u + l * 16;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)lds, 16, voff, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = 0; i < 4; ++i) out[l * 4 + i] = lds[l * 4 + i];
}
int main() {
    unsigned h[64 * 4], *d, *o, r[64 * 4];
    for (int i = 0; i < 256; ++i) h[i] = 0x1000 + i;
    hipMalloc(&d, sizeof(h)); hipMalloc(&o, sizeof(h));
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, 32 * 16, o);
    hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
    printf("lane0: %x %x | lane31: %x %x | lane32 (OOB): %x %x %x %x | lane63 (OOB): %x\n", r[0], r[1], r[31*4], r[31*4+1], r[32*4], r[32*4+1], r[32*4+2], r[32*4+3], r[63*4]);
    return 0;
}
