// Micro-experiment: which load flavours observe memory-side float atomics issued by OTHER workgroups
// (possibly other XCDs) inside one kernel, when the line was made L2-resident by an earlier kernel?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef int i4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k_fill(float *p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = 1.0f; }
__global__ void k_touch(const float *p, int n, float *sink) {   // plain loads: make lines resident in every XCD's L2
    int i = blockIdx.x * blockDim.x + threadIdx.x; float s = 0;
    for (int k = i; k < n; k += gridDim.x * blockDim.x) s += p[k];
    if (s == 123.456f) *sink = s;
}
// phase 0: every block adds 1.0 to p[0..63] (atomics). then grid-wide counter barrier. phase 1: read with flavours.
__global__ void k_test(float *p, int *counter, float *out, int nblocks) {
    const int lane = threadIdx.x;
    // make the lines resident in THIS XCD's L2 (and this CU's L1) inside the kernel, before any atomic
    float pre = p[lane] + p[64 + lane];
    if (pre == 123.0f) out[0] = pre;
    __syncthreads();
    if (lane == 0) {
        __hip_atomic_fetch_add(counter + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(counter + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < nblocks) __builtin_amdgcn_s_sleep(8);
    }
    __syncthreads();
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 0xF;
    if (xcc == 0) atomicAdd(p + lane, 1.0f);          // ONLY workgroups on XCD 0 add
    if (lane == 0) { out[(size_t)blockIdx.x * 8 * 64 + 7 * 64] = (float)xcc; if (xcc == 0) __hip_atomic_fetch_add(counter + 2, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    __syncthreads();
    if (lane == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < nblocks) __builtin_amdgcn_s_sleep(8);
    }
    __syncthreads();
    float plain = p[lane];
    float sc1 = __hip_atomic_load(p + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(p, 0, 256, 0x00020000);
    float bsc1 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, lane * 4, 0, 16));
    float bsys = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, lane * 4, 0, 17));
    float bnt  = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, lane * 4, 0, 2));
    float rmw = atomicAdd(p + 64 + lane, 0.0f);   // different words: fetch via RMW
    float rmw0 = __hip_atomic_fetch_add(p + lane, 0.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    float *o = out + (size_t)blockIdx.x * 8 * 64;
    o[0 * 64 + lane] = plain; o[1 * 64 + lane] = sc1; o[2 * 64 + lane] = bsc1; o[3 * 64 + lane] = bsys;
    o[4 * 64 + lane] = bnt; o[5 * 64 + lane] = rmw; o[6 * 64 + lane] = rmw0;
}
int main() {
    const int nblocks = 256;   // one per CU: co-resident, spins are safe
    float *p, *out, *sink; int *counter;
    CK(hipMalloc(&p, 4096)); CK(hipMalloc(&out, sizeof(float) * nblocks * 8 * 64)); CK(hipMalloc(&sink, 4)); CK(hipMalloc(&counter, 12));
    CK(hipMemset(counter, 0, 12));
    k_fill<<<4, 256>>>(p, 1024);
    k_touch<<<2048, 256>>>(p, 1024, sink);
    CK(hipDeviceSynchronize());
    k_test<<<nblocks, 64>>>(p, counter, out, nblocks);
    CK(hipDeviceSynchronize());
    std::vector<float> h(nblocks * 8 * 64);
    CK(hipMemcpy(h.data(), out, h.size() * 4, hipMemcpyDeviceToHost));
    const char *names[7] = {"plain", "atomic_load(sc1)", "buffer sc1", "buffer sc0 sc1", "buffer nt", "rmw(other words)", "rmw(+0) same"};
    int hcnt[3]; CK(hipMemcpy(hcnt, counter, 12, hipMemcpyDeviceToHost));
    const float expect = 1.0f + hcnt[2];
    printf("adders (blocks on XCD 0): %d\n", hcnt[2]);
    for (int f = 0; f < 7; ++f) {
        int fresh = 0, total = 0; float mn = 1e30f, mx = -1e30f;
        for (int b = 0; b < nblocks; ++b) { if (h[(size_t)b * 8 * 64 + 7 * 64] == 0.0f) continue;   // readers on the OTHER XCDs only
          for (int l = 0; l < 64; ++l) {
            float v = h[(size_t)b * 8 * 64 + f * 64 + l]; total++; if (v == expect) fresh++; if (v < mn) mn = v; if (v > mx) mx = v;
        } }
        printf("%-20s fresh %5d / %5d   min %g max %g (expect %g)\n", names[f], fresh, total, mn, mx, expect);
    }
    return 0;
}
