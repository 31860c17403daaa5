// exp_tr_read.hip -- probe the semantics of ds_read_b64_tr_b16 on gfx950.
// LDS is filled with 16-bit values equal to their own element index; every lane supplies an arbitrary byte address;
// the four 16-bit results per lane are dumped so that the data movement between lanes can be read off.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__global__ void probe(const int* lane_addr_bytes, uint16_t* out) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
    __syncthreads();
    const unsigned base = (unsigned)(uintptr_t)(&lds[0]);      // LDS byte address (low 32 bits of the generic pointer)
    const unsigned addr = base + (unsigned)lane_addr_bytes[threadIdx.x];
    uint2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    out[threadIdx.x * 4 + 0] = (uint16_t)(v.x & 0xffff);
    out[threadIdx.x * 4 + 1] = (uint16_t)(v.x >> 16);
    out[threadIdx.x * 4 + 2] = (uint16_t)(v.y & 0xffff);
    out[threadIdx.x * 4 + 3] = (uint16_t)(v.y >> 16);
}

static void run(const char* name, const std::vector<int>& addr) {
    int* d_a; uint16_t* d_o;
    hipMalloc(&d_a, 64 * 4); hipMalloc(&d_o, 64 * 4 * 2);
    hipMemcpy(d_a, addr.data(), 64 * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d_a, d_o);
    std::vector<uint16_t> o(256);
    hipMemcpy(o.data(), d_o, 512, hipMemcpyDeviceToHost);
    printf("== %s (err %d)\n", name, (int)hipGetLastError());
    for (int l = 0; l < 64; ++l) {
        printf("lane %2d addr(elem) %4d -> %4d %4d %4d %4d\n", l, addr[l] / 2, o[l * 4], o[l * 4 + 1], o[l * 4 + 2], o[l * 4 + 3]);
    }
}

int main() {
    std::vector<int> a(64);
    // case 1: every lane reads "its" natural 8-byte chunk: lane l -> element 4*l (contiguous 512 B)
    for (int l = 0; l < 64; ++l) a[l] = l * 8;
    run("natural contiguous chunks (lane l -> elem 4l)", a);
    // case 2: a row-major [k][n] matrix with row stride 136 elements; 16-lane group g handles cols 16*(g&1).. and rows 4*(g>>1)..:
    //         lane c of a group supplies the chunk of row c/4, cols 4*(c%4)
    for (int l = 0; l < 64; ++l) { int g = l >> 4, c = l & 15; a[l] = (((g >> 1) * 4 + c / 4) * 136 + (g & 1) * 16 + 4 * (c % 4)) * 2; }
    run("4x16 blocks of a stride-136 matrix", a);
    // case 3: all lanes the same address
    for (int l = 0; l < 64; ++l) a[l] = 64;
    run("uniform address (elem 32)", a);
    return 0;
}
