// Cost of a grid-wide barrier inside a persistent kernel (one workgroup per CU, agent-scope counter + polling) against the
// ~6 us gap between two dependent launches: is a one-launch decode step worth building?
// hipcc --offload-arch=gfx950 -O3 scripts/ubench/grid_barrier.hip -o scripts/ubench/grid_barrier
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

__device__ __forceinline__ void st_agent(float* p, float v) {
    __hip_atomic_store(reinterpret_cast<unsigned*>(p), __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float ld_agent(const float* p) {
    return __uint_as_float(__hip_atomic_load(reinterpret_cast<const unsigned*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
// every wave: own stores acknowledged; one lane: arrive + poll (bounded: a lost workgroup must not hang the box)
__device__ __forceinline__ bool grid_sync(unsigned* ctr, unsigned target, int* err) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const long long t0 = wall_clock64();
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (wall_clock64() - t0 > 100000000ll) { *err = 1; break; }      // 1 s at 100 MHz
        }
    }
    __syncthreads();
    return true;
}

// phases: every workgroup writes a value derived from the phase, syncs, reads the value of workgroup (id + 37) % G
template <int WORK>
__global__ __launch_bounds__(256) void k(unsigned* ctr, float* buf, int phases, int* err, float* out) {
    const int G = gridDim.x, wg = blockIdx.x, tid = threadIdx.x;
    float acc = 0.f;
    for (int p = 0; p < phases; ++p) {
        st_agent(buf + (size_t)(p & 1) * G * 256 + wg * 256 + tid, (float)(p * 3 + wg + tid));
        grid_sync(ctr, (unsigned)(p + 1) * G, err);
        const int o = (wg + 37) % G;
        const float v = ld_agent(buf + (size_t)(p & 1) * G * 256 + o * 256 + tid);
        if (v != (float)(p * 3 + o + tid)) *err = 2;
        acc += v;
        if (WORK) {                     // some dependent arithmetic between barriers
            for (int i = 0; i < WORK; ++i) acc = acc * 1.0001f + 0.5f;
        }
    }
    out[wg * 256 + tid] = acc;
}

int main(int argc, char** argv) {
    unsigned* ctr; float *buf, *out; int* err;
    (void)hipMalloc(&ctr, 256); (void)hipMalloc(&buf, 2 * 1024 * 256 * 4); (void)hipMalloc(&out, 1024 * 256 * 4); (void)hipMalloc(&err, 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int phases = 2000;
    for (int G : {32, 64, 128, 256}) {
        float best = 1e9; int herr = 0;
        for (int rep = 0; rep < 3; ++rep) {
            (void)hipMemset(ctr, 0, 256); (void)hipMemset(err, 0, 4);
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(k<0>, dim3(G), dim3(256), 0, 0, ctr, buf, phases, err, out);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
            int h; (void)hipMemcpy(&h, err, 4, hipMemcpyDeviceToHost); herr |= h;
        }
        printf("G=%3d workgroups: %.2f us per phase (store + grid barrier + load), err %d\n", G, best * 1e3 / phases, herr);
    }
    return 0;
}
