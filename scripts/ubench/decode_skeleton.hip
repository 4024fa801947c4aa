// Skeleton of a ONE-LAUNCH decode step (round 5): a persistent kernel of G workgroups walks the 34 phases of
// tal_greedy_step_fwd (embed, 4 x {qkv, self-attention, out-proj, cross q, cross-attention, out-proj, FFN-1, FFN-2}, LM head + pick)
// behind grid barriers.  It streams the REAL operand bytes of every phase (weights: requested BEFORE the workgroup arrives at the
// barrier that guards the phase's activations; activations: after it) and publishes real-sized outputs (write-through stores), with
// an optional stand-in for the phase's MFMA time.  No results: this prices the structure before it is built.
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench/decode_skeleton.hip -o scripts/ubench/decode_skeleton
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct Phase {
    long long w_ofs;      // floats, into the weight arena
    int w_bytes;          // static operand bytes of the phase (whole phase, all workgroups)
    int units;            // work units (column blocks / (head, chunk) pairs): a unit's static bytes = w_bytes / units
    int a_bytes;          // activation bytes EVERY active workgroup reads after the barrier
    int o_bytes;          // bytes a unit publishes
    int mfma;             // 16x16x4 fp32 MFMAs per wave and unit (stand-in for the arithmetic)
};
constexpr int MAXP = 40;
struct Plan {
    Phase p[MAXP];
    int n;
};

// arrive: every wave's published stores are acknowledged, one lane adds; wait: that lane polls.  Between the two the workgroup requests
// the NEXT phase's static operands (they do not depend on anybody's outputs), so their round trip runs under the wait.
// LOCAL: every participant sits on ONE XCD (checked by the caller through XCC_ID): the counter lives in that XCD's L2 (an atomic without
// sc1 executes there), the poll and the payload loads only have to miss the CU's L1 (sc0), the payload stores are plain (the L1 writes through)
template <bool LOCAL>
__device__ __forceinline__ void grid_arrive(unsigned* ctr) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        if (LOCAL) asm volatile("global_atomic_add %0, %1, off" ::"v"(ctr), "v"(1u) : "memory");
        else __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
template <bool LOCAL>
__device__ __forceinline__ unsigned poll_word(unsigned* ctr) {
    if (!LOCAL) return __hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned v;      // a returning atomic executes in the L2 whatever the L1 holds
    asm volatile("global_atomic_or %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(ctr), "v"(0u) : "memory");
    return v;
}
template <bool LOCAL>
__device__ __forceinline__ void grid_wait(unsigned* ctr, unsigned target, int* err) {
    if (threadIdx.x == 0) {
        const long long t0 = wall_clock64();
        while (poll_word<LOCAL>(ctr) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (wall_clock64() - t0 > 200000ll || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 1) {      // 2 ms; one give-up ends the launch
                __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
    }
    __syncthreads();
}

// agent-scope (sc1: aux bit 4) 16-byte loads / write-through stores through a buffer descriptor: the compiler counts them (an asm load's
// destination registers are fair game for the register allocator while the load is still in flight)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t mk_rsrc(const float* p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, 0x7fffffff, 0x00020000);
}
template <int AUX>
__device__ __forceinline__ f32x4 ld16_sc1(__amdgpu_buffer_rsrc_t r, int byte_off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, AUX));
}
template <int AUX>
__device__ __forceinline__ void st16_sc1(__amdgpu_buffer_rsrc_t r, int byte_off, f32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, byte_off, 0, AUX);
}

// AMODE 0: activations by agent-scope (sc1) 16-byte loads; 1: one acquire fence (buffer_inv sc1) after the barrier, plain loads;
// 2: the one-XCD protocol (see grid_arrive)
// PLACE 0: workgroup b works; 1: only workgroups with b % 8 == 0 work (one XCD under round-robin dispatch), index b / 8
template <int T, int AMODE, int WREG>
__global__ __launch_bounds__(T) void skel(const Plan plan, const float* __restrict__ warena, float* act, unsigned* ctr, int* err,
                                          float* sink, int G, int place, long long* tstamp, int flags) {
    int wg = blockIdx.x;
    if (place == 1) {
        if (wg & 7) return;
        wg >>= 3;
    }
    const int tid = threadIdx.x;
    float acc = 0.f;
    f32x4 wreg[WREG];
    auto prefetch = [&](int ph) {
        const Phase& P = plan.p[ph];
        // this workgroup's share of the phase's static operand bytes (capped by the registers that hold them: WREG x T x 16 bytes)
        const int share = (P.w_bytes / G) & ~15;
        const float* base = warena + P.w_ofs + (long long)wg * (share / 4);
#pragma unroll
        for (int i = 0; i < WREG; ++i) {
            const int o = (i * T + tid) * 16;
            if (o < share && !(flags & 1)) wreg[i] = *reinterpret_cast<const f32x4*>(base + o / 4);
            else wreg[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    prefetch(0);
    const bool tsw = tstamp && tid == 0 && (wg == 0 || wg == G - 1);
    long long* ts = tstamp + (wg == 0 ? 0 : 5 * MAXP);
    for (int ph = 0; ph < plan.n; ++ph) {
        const Phase& P = plan.p[ph];
        // ---- wait for the previous phase's outputs
        if (ph > 0) grid_wait<AMODE == 2>(ctr, (unsigned)ph * (unsigned)G, err);
        if (tsw) ts[5 * ph] = wall_clock64();
        if (AMODE == 2) {
            asm volatile("buffer_inv sc0" ::: "memory");          // drop the CU's L1 lines: the loads below come from the L2
        }
        if (AMODE == 1 && ph > 0) {
            if (tid == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            __syncthreads();
        }
        if (ph > 0 && tid < G) {
            const __amdgpu_buffer_rsrc_t rc = mk_rsrc(act + (size_t)(2 << 20) + (size_t)(ph & 1) * 4096);
            f32x4 c;
            if (AMODE == 0) c = ld16_sc1<16>(rc, tid * 16);
            else c = ld16_sc1<0>(rc, tid * 16);
            if (c.x != (float)ph || c.w != (float)ph) *err = 2;
        }
        const int my_units = wg < P.units ? (P.units - wg + G - 1) / G : 0;
        if (my_units > 0 && P.a_bytes > 0 && !(flags & 2)) {
            const float* a = act + (size_t)(ph & 1) * (1 << 20);
            const __amdgpu_buffer_rsrc_t ra = mk_rsrc(a);
            // 16 bytes per lane and load, up to AL loads in flight per lane (64 KB per workgroup in one round trip)
            constexpr int AL = 16 * 256 / T;
            const int n16 = P.a_bytes / 16;
            for (int i0 = tid; i0 < n16; i0 += AL * T) {
                f32x4 v[AL];
#pragma unroll
                for (int k = 0; k < AL; ++k) {
                    const int i = i0 + k * T;
                    const float* p = a + (size_t)(i < n16 ? i : 0) * 4;
                    if (AMODE == 0) v[k] = ld16_sc1<16>(ra, (i < n16 ? i : 0) * 16);

                    else v[k] = *reinterpret_cast<const f32x4*>(p);
                }
#pragma unroll
                for (int k = 0; k < AL; ++k) acc += v[k].x + v[k].y + v[k].z + v[k].w;
            }
        }
        if (tsw) { asm volatile("" : "+v"(acc)); ts[5 * ph + 1] = wall_clock64(); }
        // ---- consume the static operands (requested before the barrier), stand-in arithmetic
#pragma unroll
        for (int i = 0; i < WREG; ++i) acc += wreg[i].x + wreg[i].y + wreg[i].z + wreg[i].w;
        if (tsw) { asm volatile("" : "+v"(acc)); ts[5 * ph + 2] = wall_clock64(); }
        if (my_units > 0 && P.mfma > 0) {
            f32x4 c0 = {acc, 0.f, 0.f, 0.f}, c1 = c0;
            const int n = P.mfma * my_units / 2;
            for (int i = 0; i < n; ++i) {
                c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(acc, 1.0f, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(acc, 1.0f, c1, 0, 0, 0);
            }
            acc += (c0.x + c1.x) * 1e-30f;
        }
        if (tsw) { asm volatile("" : "+v"(acc)); ts[5 * ph + 3] = wall_clock64(); }
        // ---- publish, arrive, and only then request the next phase's static operands: in flight while the barrier fills
        if (my_units > 0) {
            float* o = act + (size_t)((ph + 1) & 1) * (1 << 20);
            const int n16 = P.o_bytes * my_units / 16;
            const __amdgpu_buffer_rsrc_t ro = mk_rsrc(o);
            for (int i = tid; i < n16; i += T) {
                if (AMODE == 2) st16_sc1<0>(ro, (int)(((size_t)wg * n16 + i) % (1 << 18)) * 16, f32x4{acc, acc, acc, acc});
                else st16_sc1<16>(ro, (int)(((size_t)wg * n16 + i) % (1 << 18)) * 16, f32x4{acc, acc, acc, acc});
            }
        }
        if (tid == 0) {
            const __amdgpu_buffer_rsrc_t rc = mk_rsrc(act + (size_t)(2 << 20) + (size_t)((ph + 1) & 1) * 4096);
            const float t = (float)(ph + 1);
            if (AMODE == 2) st16_sc1<0>(rc, wg * 16, f32x4{t, t, t, t});
            else st16_sc1<16>(rc, wg * 16, f32x4{t, t, t, t});
        }
        if (ph + 1 < plan.n) {
            if (flags & 4) prefetch(ph + 1);          // (the first skeleton's order: requested before the arrive, which then waits for them)
            grid_arrive<AMODE == 2>(ctr);
            if (tsw) ts[5 * ph + 4] = wall_clock64();
            if (!(flags & 4)) prefetch(ph + 1);
        }
    }
    sink[(size_t)wg * T + tid] = acc;
}

static Plan make_plan(int U, int S, int E, int FF, int H, int V, int E0, int L, bool fold, long long* wfloats) {
    Plan pl;
    memset(&pl, 0, sizeof(pl));
    int n = 0;
    long long ofs = 0;
    const int rows = (U + 15) / 16 * 16;
    const int mt = rows / 16;
    auto add = [&](long long wbytes, int units, int abytes, int obytes, int mfma) {
        Phase& p = pl.p[n++];
        p.w_ofs = ofs;
        p.w_bytes = (int)wbytes;
        p.units = units;
        p.a_bytes = abytes;
        p.o_bytes = obytes;
        p.mfma = mfma;
        ofs += (wbytes / 4 + 1023) / 1024 * 1024;
    };
    const int act = U * E * 4;
    auto gemm = [&](int N, int K, int ksplit) {      // N / 16 column blocks x ksplit K slices, 4 waves split K
        add((long long)N * K * 4, N / 16 * ksplit, U * K * 4, U * 16 * 4, mt * (K / ksplit / 4 / 4));
    };
    add((long long)U * E0 * 4 + (long long)E * E0 * 4, U, U * 8, E * 4, 0);                    // embed
    for (int l = 0; l < L; ++l) {
        gemm(3 * E, E, 1);                                                                    // q | k | v^T
        add(0, H * mt, U * 3 * (E / H) * 4, U * (E / H) * 4, mt * (2 * (E / H) / 4));         // self-attention
        gemm(E, E, 1);                                                                        // out-proj + rezero
        if (!fold) {
            gemm(E, E, 1);                                                                    // cross q
            add((long long)2 * S * E * 4, H * 8, U * (E / H) * 4, U * (E / H + 4) * 4 + 16 * 48 * 4, mt * 16);   // key-split cross-attention
            gemm(E, E, 1);                                                                    // out-proj + rezero
        } else {
            // q-projection folded into K, out-projection into V: scores = x1 . K'^T (K = E), x2 = x1 + rw (P . V' + b) (N = E per head)
            add((long long)2 * S * E * H * 4, H * 8, act, U * (E + 4) * 4 + 16 * 48 * 4, mt * (E / 4 / 4 + 4 * 3 * (E / 16) / 4));
        }
        gemm(FF, E, 1);                                                                       // FFN-1
        gemm(E, FF, 4);                                                                       // FFN-2, K cut over 4 workgroups
    }
    add((long long)V * E0 * 4 + (long long)E * E0 * 4, (V + 127) / 128, E * 4, 8, 0);         // LM head + pick
    pl.n = n;
    *wfloats = ofs;
    return pl;
}

template <int T, int AMODE, int WREG>
static float run(const Plan& pl, const float* w, float* act, unsigned* ctr, int* err, float* sink, int G, int place, int flags, int* herr,
                 long long* tstamp) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    float best = 1e9f;
    const int reps = 5;
    for (int rep = 0; rep < reps; ++rep) {
        (void)hipMemsetAsync(ctr, 0, 256, 0);
        (void)hipMemsetAsync(err, 0, 4, 0);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((skel<T, AMODE, WREG>), dim3(place == 1 ? 8 * G : G), dim3(T), 0, 0, pl, w, act, ctr, err, sink, G, place,
                           rep == reps - 1 ? tstamp : nullptr, flags);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
        int h;
        (void)hipMemcpy(&h, err, 4, hipMemcpyDeviceToHost);
        *herr |= h;
    }
    return best;
}

int main(int argc, char** argv) {
    const int U = argc > 1 ? atoi(argv[1]) : 20;
    const int S = 357, E = 512, FF = 2048, H = 4, V = 10000, E0 = 64, L = 4;
    unsigned* ctr;
    int* err;
    float *act, *sink, *w;
    long long* ts;
    (void)hipMalloc(&ctr, 256);
    (void)hipMalloc(&err, 4);
    (void)hipMalloc(&act, (size_t)3 * (1 << 20) * 4);
    (void)hipMalloc(&sink, (size_t)2048 * 1024 * 4);
    (void)hipMalloc(&ts, 10 * MAXP * 8);
    (void)hipMemset(act, 0, (size_t)3 * (1 << 20) * 4);
    for (int fold = 0; fold < 2; ++fold) {
        long long wf = 0;
        Plan pl = make_plan(U, S, E, FF, H, V, E0, L, fold != 0, &wf);
        (void)hipMalloc(&w, (size_t)wf * 4);
        (void)hipMemset(w, 0, (size_t)wf * 4);
        long long wb = 0;
        for (int i = 0; i < pl.n; ++i) wb += pl.p[i].w_bytes;
        printf("prefix %d tokens, %d phases%s, %.1f MB of static operands per step\n", U, pl.n, fold ? " (cross q / out projections folded into K / V)" : "",
               wb / 1e6);
        // flags: 1 = no static operands, 2 = no activation loads, 4 = static operands requested before the arrive, 8 = no arithmetic
        const int fl[] = {0, 4, 8, 8 | 1, 8 | 2, 8 | 1 | 2};
        const char* fn[] = {"all                          ", "all, operands before arrive  ", "no math                      ", "no math, no static operands  ",
                            "no math, no activations      ", "barriers + publishing only   "};
        for (int fi = 0; fi < 6; ++fi) {
            Plan q = pl;
            if (fl[fi] & 8) for (int i = 0; i < q.n; ++i) q.p[i].mfma = 0;
            for (int place = 0; place < 2; ++place)
                for (int G : {32, 64, 128}) {
                    if (place == 1 && G > 32) continue;
                    int herr = 0;
                    float t[4];
                    t[0] = run<256, 0, 32>(q, w, act, ctr, err, sink, G, place, fl[fi], &herr, ts);
                    std::vector<long long> hts(10 * MAXP);
                    (void)hipMemcpy(hts.data(), ts, 10 * MAXP * 8, hipMemcpyDeviceToHost);
                    t[1] = run<256, 1, 32>(q, w, act, ctr, err, sink, G, place, fl[fi], &herr, nullptr);
                    t[2] = run<512, 0, 16>(q, w, act, ctr, err, sink, G, place, fl[fi], &herr, nullptr);
                    t[3] = run<512, 1, 16>(q, w, act, ctr, err, sink, G, place, fl[fi], &herr, nullptr);
                    printf("  %s G=%3d %s: 256 thr sc1-loads %.1f us | acquire+plain %.1f us | 512 thr sc1-loads %.1f us | acquire+plain %.1f us  (err %d)\n",
                           fn[fi], G, place ? "one XCD" : "spread ", t[0] * 1e3, t[1] * 1e3, t[2] * 1e3, t[3] * 1e3, herr);
                    if (place == 1) {
                        const float a = run<256, 2, 32>(q, w, act, ctr, err, sink, G, place, fl[fi], &herr, ts);
                        (void)hipMemcpy(hts.data(), ts, 10 * MAXP * 8, hipMemcpyDeviceToHost);
                        const float b = run<512, 2, 16>(q, w, act, ctr, err, sink, G, place, fl[fi], &herr, nullptr);
                        fflush(stdout);
                        printf("  %s G=%3d one XCD, hand-offs through its L2: 256 thr %.1f us | 512 thr %.1f us  (err %d)\n", fn[fi], G, a * 1e3, b * 1e3, herr);
                    }
                    fflush(stdout);
                    if (fi == 0) {
                        for (int who = 0; who < 2; ++who) {
                            const long long* h = hts.data() + who * 5 * MAXP;
                            double s[5] = {0, 0, 0, 0, 0};
                            for (int i = 0; i + 1 < q.n; ++i) {
                                s[0] += (h[5 * i + 1] - h[5 * i]) / 100.0;           // activations loaded
                                s[1] += (h[5 * i + 2] - h[5 * i + 1]) / 100.0;       // static operands consumed
                                s[2] += (h[5 * i + 3] - h[5 * i + 2]) / 100.0;       // arithmetic
                                s[3] += (h[5 * i + 4] - h[5 * i + 3]) / 100.0;       // publish + arrive
                                s[4] += (h[5 * (i + 1)] - h[5 * i + 4]) / 100.0;     // request next operands + wait
                            }
                            printf("    workgroup %3d, us per step: activations %.1f | static operands landed %.1f | arithmetic %.1f | publish + arrive %.1f | wait %.1f\n",
                                   who ? G - 1 : 0, s[0], s[1], s[2], s[3], s[4]);
                        }
                    }
                }
        }
        (void)hipFree(w);
    }
    return 0;
}
