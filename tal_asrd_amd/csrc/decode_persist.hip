// The decode step of System.generate_unaligned (tal/asr/system.py:332-411) as ONE launch: G persistent workgroups per session walk
// the step's phases -- embed, per decoder layer {q | k | v^T, self-attention, out-projection + ReZero, cross q, key-split
// cross-attention, out-projection + ReZero, FFN-1, FFN-2 cut along K}, LM head + pick -- and call, for the blocks dealt to them, the
// SAME device bodies the launch chain's kernels run (csrc/decode_bodies.h) with the same arguments: every token, attention row and
// hidden state is bit-identical to tal_greedy_step_fwd's chain of 34 launches.  What changes is what sits between two phases: a
// counter barrier among the session's workgroups instead of a kernel boundary.
//
// Hand-off protocol (the MI355X guide's recipe R1, priced in profiles/r5_decode_persistent_skeleton.txt): a phase's outputs are
// stored write-through at agent scope (the bodies' COH form); every wave drains its stores (s_waitcnt vmcnt(0)), the workgroup
// meets at its own barrier, ONE lane adds 1 to the session's arrival counter (agent scope, relaxed) and polls it (relaxed load +
// s_sleep) until phase x G workgroups have arrived; then ONE lane issues an agent-scope acquire fence (the CU's L1 and the L2's
// non-local lines are dropped) and the workgroup's plain loads of the next phase are fresh.  Counters live in the session's ticket
// block (words PS_BAR, PS_DONE; PS_ERR is raised when a wait gives up) and are left at zero by the launch, like the other tickets.
// Sessions are independent: several of them (the decode sessions of System.transcribe_unaligned_many) run side by side in one
// launch, G workgroups each, without ever waiting for one another.
#include "decode_bodies.h"

namespace tal {

// (host-side mirror in decoder.hip fills these; everything by value in the kernel argument segment: <= 4 KB)
__device__ __forceinline__ SkinnyArgs ps_skinny_args(const float* A, int64_t lda, const float* W, const float* bias, const float* res, float* Y,
                                                     int64_t ldy, int M, int N, int K, float alpha) {
    SkinnyArgs g = {};
    g.A = A; g.W = W; g.bias = bias; g.res = res; g.Y = Y;
    g.M = M; g.N = N; g.K = K;
    g.lda = lda; g.ldw = K; g.ldy = ldy; g.ldres = ldy;
    g.alpha = alpha;
    return g;
}

// (ablation build -DPS_TIMELINE: workgroups 0 and G - 1 of session 0 stamp the 100 MHz wall clock around every phase barrier;
//  scripts/decode_persist_timeline.py reads them through tal_debug_ps_timeline.  Not part of the product library.)
#ifdef PS_TIMELINE
__device__ unsigned long long g_ps_tl[2 * 64 * 4];
#define PS_STAMP(b, i)                                                                                         \
    do {                                                                                                       \
        if (threadIdx.x == 0 && (b).tl >= 0 && (b).phase < 64) g_ps_tl[((b).tl * 64 + (b).phase) * 4 + (i)] = wall_clock64(); \
    } while (0)
#else
#define PS_STAMP(b, i)
#endif

struct PsBarrier {
    unsigned* bar;
    unsigned* err;
    unsigned G, phase;
    int tl;
    float* out;              // the session's result buffer {token, row [S] (, sequence word)}: receives the failure marker
    int S;
    unsigned host_seq;
    bool wg0;                // the session's first workgroup: writes the marker whenever it sees the step fail
};
// A step that cannot finish (a barrier wait gave up: its workgroups were not resident together, or one of them died) must not look
// like a step that did: the FIRST workgroup to raise the error word writes the token -1 and, for a host-polled step, the sequence
// word, so that the host's poll returns and finds the marker (tal_greedy_step_fwd / tal_greedy_step_poll then report TAL_EHIP and
// the context's ticket block -- phase counters, error word, half-counted attention / split-K tickets -- is zeroed before its next
// step).  Every workgroup checks the error word before, inside and after its wait, so none runs a phase behind a failed barrier;
// the session's first workgroup writes the marker too when it finds the word raised (a word left raised by an earlier failure
// has no first raiser in this launch).
__device__ __forceinline__ void ps_mark(PsBarrier& b) {
    __hip_atomic_store(reinterpret_cast<unsigned*>(b.out), 0xffffffffu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (b.host_seq) __hip_atomic_store(reinterpret_cast<unsigned*>(b.out + 1 + b.S), b.host_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void ps_raise(PsBarrier& b) {
    if (__hip_atomic_exchange(b.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) ps_mark(b);      // (the first to raise it)
}
// -> false: a wait gave up (or another workgroup's did): the caller leaves the kernel without delivering a result
__device__ __forceinline__ bool ps_sync(PsBarrier& b) {
    __shared__ unsigned ok_sh;
    PS_STAMP(b, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // every wave: its write-through stores are acknowledged
    __syncthreads();
    PS_STAMP(b, 1);
    if (threadIdx.x == 0) {
        ++b.phase;
        __hip_atomic_fetch_add(b.bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned target = b.phase * b.G;
        const long long t0 = wall_clock64();
        // (the error word is looked at before the wait too: a counter left at or beyond the target by a step that failed earlier
        //  would let every wait fall through without ordering anything)
        unsigned ok = __hip_atomic_load(b.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u ? 1u : 0u;
        while (ok && __hip_atomic_load(b.bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (wall_clock64() - t0 > 20000000ll) {                                     // 200 ms at 100 MHz
                ps_raise(b);
                ok = 0u;
            } else if (__hip_atomic_load(b.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
                ok = 0u;
            }
        }
        if (ok && __hip_atomic_load(b.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) ok = 0u;
        if (!ok && b.wg0) ps_mark(b);
#ifdef PS_TIMELINE
        --b.phase; PS_STAMP(b, 2); ++b.phase;
#endif
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        ok_sh = ok;
    }
    __syncthreads();
#ifdef PS_TIMELINE
    --b.phase; PS_STAMP(b, 3); ++b.phase;
#endif
    return ok_sh != 0u;
}

template <int MODE>
__device__ __forceinline__ void ps_skinny(const SkinnyArgs& g, unsigned wg, unsigned G) {
    const unsigned gx = (unsigned)(g.N / 16), gy = (unsigned)((g.M + 31) / 32), gz = (unsigned)(g.ksplit > 1 ? g.ksplit : 1);
    for (unsigned vb = wg; vb < gx * gy * gz; vb += G) {
        const Blk blk{vb % gx, (vb / gx) % gy, vb / (gx * gy), gx, gy};
        if (g.M <= 16) skinny_gemm_body<MODE, 1, 4, true>(g, blk);
        else skinny_gemm_body<MODE, 2, 4, true>(g, blk);
        __syncthreads();          // (the body's LDS buffers are reused by the next block)
    }
}

template <int HD>
__global__ __launch_bounds__(256) void greedy_persist_kernel(const PsArgs a) {
    const unsigned G = (unsigned)a.G;
    const unsigned si = blockIdx.x / G, wg = blockIdx.x - si * G;
    const PsSession& s = a.s[si];
    const PsModel& m = a.m;
    const int E = m.E, H = m.H, FF = m.FF, U = s.U, S = s.S, K0 = m.K0;
    const int64_t U4 = (U + 3) & ~3, S4 = (S + 3) & ~3;
    const float qscale = m.qscale;
    PsBarrier bar{s.tickets + PS_BAR, s.tickets + PS_ERR, G, 0u, si == 0 ? (wg == 0 ? 0 : (wg == G - 1 ? 1 : -1)) : -1, s.out, S, s.host_seq, wg == 0};

    // ---- embed (models.py:218-223)
    for (unsigned row = wg; row < (unsigned)U; row += G) {
        embed_body<true>(s.tokens, m.emb, m.proj, m.pe, s.h0, U, m.V, K0, E, (int)row);
        __syncthreads();
    }
    if (!ps_sync(bar)) return;

    const float* cur = s.h0;
    for (int l = 0; l < m.n_layers; ++l) {
        const tal_decoder_layer_w& w = m.layer[l];
        // q | k | v^T of the self-attention (q scaled, V stored transposed with its bias)
        {
            SkinnyArgs g = ps_skinny_args(cur, E, w.sa_in_w, w.sa_in_b, nullptr, s.qkv, 3 * E, U, 3 * E, E, qscale);
            g.scale_cols = E;
            g.Yt = s.vt; g.vt_begin = 2 * E; g.U = U; g.ldt = U4; g.vt_bs = (int64_t)E * U4;
            ps_skinny<3>(g, wg, G);
        }
        if (!ps_sync(bar)) return;
        {
            AttnArgs t = {};
            t.q = s.qkv; t.ldq = 3 * E; t.q_bs = (int64_t)U * 3 * E;
            t.k = s.qkv + E; t.ldk = 3 * E; t.k_bs = (int64_t)U * 3 * E;
            t.vt = s.vt; t.ldvt = U4; t.vt_bs = (int64_t)E * U4;
            t.ctx = s.ctx; t.ldc = E; t.c_bs = (int64_t)U * E;
            t.U = U; t.S = U; t.H = H;
            const unsigned gx = (unsigned)((U + 15) / 16);
            for (unsigned vb = wg; vb < gx * (unsigned)H; vb += G) {
                attn_small_body<HD, 4, true>(t, Blk{vb % gx, vb / gx, 0u, gx, (unsigned)H});
                __syncthreads();
            }
        }
        if (!ps_sync(bar)) return;
        ps_skinny<2>(ps_skinny_args(s.ctx, E, w.sa_out_w, w.sa_out_b, cur, s.x1, E, U, E, E, w.resweight), wg, G);
        if (!ps_sync(bar)) return;
        // cross attention over the cached K / V^T of the encoder window
        ps_skinny<3>(ps_skinny_args(s.x1, E, w.ca_in_w, w.ca_in_b, nullptr, s.qkv, E, U, E, E, qscale), wg, G);
        if (!ps_sync(bar)) return;
        {
            AttnArgs c = {};
            c.q = s.qkv; c.ldq = E; c.q_bs = (int64_t)U * E;
            c.k = s.k_cache[l]; c.ldk = s.k_pitch ? s.k_pitch : E; c.k_bs = (int64_t)S * c.ldk;
            c.vt = s.vt_cache[l]; c.ldvt = S4; c.vt_bs = (int64_t)E * S4;
            c.vbias = w.ca_in_b + 2 * E;
            c.kpm = s.kpm;
            c.ctx = s.ctx; c.ldc = E; c.c_bs = (int64_t)U * E;
            c.U = U; c.S = S; c.H = H;
            c.probs = s.probs + (size_t)l * H * S; c.prob_row0 = U - 1;
            const int CB = split_cb(S), nblk = (S + 15) / 16, NCH = (nblk + CB - 1) / CB;
            const unsigned gx = (unsigned)((U + 15) / 16);
            for (unsigned vb = wg; vb < gx * (unsigned)H * (unsigned)NCH; vb += G) {
                attn_split_body<HD, true>(c, CB, NCH, s.sk_part, s.tickets, Blk{vb % gx, (vb / gx) % (unsigned)H, vb / (gx * (unsigned)H), gx, (unsigned)H});
                __syncthreads();
            }
        }
        if (!ps_sync(bar)) return;
        ps_skinny<2>(ps_skinny_args(s.ctx, E, w.ca_out_w, w.ca_out_b, s.x1, s.x2, E, U, E, E, w.resweight_src), wg, G);
        if (!ps_sync(bar)) return;
        // feed-forward
        ps_skinny<1>(ps_skinny_args(s.x2, E, w.lin1_w, w.lin1_b, nullptr, s.ff, FF, U, FF, E, 0.f), wg, G);
        if (!ps_sync(bar)) return;
        {
            SkinnyArgs f2 = ps_skinny_args(s.ff, FF, w.lin2_w, w.lin2_b, s.x2, s.h1, E, U, E, FF, w.resweight);
            f2.ksplit = 4;
            f2.sk_part = s.sk_part;
            f2.sk_tickets = s.tickets + 64;
            ps_skinny<2>(f2, wg, G);
        }
        if (!ps_sync(bar)) return;
        cur = s.h1;
    }
    // ---- tied factorised LM head on the last position + pick + the new token's attention row (models.py:243-246, system.py:355-411)
    {
        LmPickArgs q = {};
        q.h = cur + (size_t)(U - 1) * E;
        q.attn = s.probs;
        q.layer_stride = (int64_t)H * S;
        q.head_stride = (int64_t)S;
        q.S = S;
        q.partial = s.pick_part;
        q.ticket_word = s.tickets + (TAL_GREEDY_TICKETS - 1);
        q.out = s.out;
        q.token_out = s.token_out;
        q.host_seq = s.host_seq;
        q.bias = nullptr;          // (a context with an LM row does not take the one-launch form: greedy_persist_ok)
        const unsigned gx = (unsigned)((m.V + LMP_ROWS - 1) / LMP_ROWS);
        for (unsigned bx = wg; bx < gx; bx += G) {
            lm_pick_body(q, m.proj_t, E, K0, m.emb, m.V, m.n_layers, H, bx, gx);
            __syncthreads();
        }
    }
    // ---- the counters go back to zero behind the last workgroup (every workgroup is past every wait by then)
    if (threadIdx.x == 0) {
        const unsigned d = __hip_atomic_fetch_add(s.tickets + PS_DONE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (d == G - 1) {
            __hip_atomic_store(s.tickets + PS_BAR, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(s.tickets + PS_DONE, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// dynamic LDS of the launch: the largest of its phases' (embedding row | self-attention scores | cross-attention chunk | pick)
static size_t ps_lds_bytes(const PsArgs& a) {
    size_t n = (size_t)a.m.K0;
    const size_t pick = (size_t)a.m.E + a.m.K0 + LMP_ROWS;
    n = pick > n ? pick : n;
    for (int i = 0; i < a.n; ++i) {
        const size_t self = (size_t)16 * (((a.s[i].U + 15) & ~15) + 4);
        const size_t cross = (size_t)16 * (split_cb(a.s[i].S) * 16 + 4);
        n = self > n ? self : n;
        n = cross > n ? cross : n;
    }
    return n * sizeof(float);
}

int launch_greedy_persist(const PsArgs& a, hipStream_t s) {
    TAL_CHECK_ARG(a.n >= 1 && a.n <= TAL_PS_MAX_SESS && a.G >= 1, "one-launch decode step: %d sessions x %d workgroups", a.n, a.G);
    TAL_CHECK_ARG(a.n * a.G <= device_cus(), "one-launch decode step: %d workgroups must be resident together on %d CUs", a.n * a.G, device_cus());
    const int hd = a.m.E / a.m.H;
    const size_t lds = ps_lds_bytes(a);
    TAL_CHECK_ARG(lds <= 48 * 1024, "one-launch decode step: %zu bytes of LDS", lds);
    const dim3 grid((unsigned)(a.n * a.G));
    double work = 0.0;
    for (int i = 0; i < a.n; ++i) work += 2.0 * a.s[i].U * (double)a.m.n_layers * ((double)a.m.E * (6.0 * a.m.E + 2.0 * a.m.FF));
    ProfScope prof(PROF_OTHER, work, s);
    switch (hd) {
        case 128: hipLaunchKernelGGL((greedy_persist_kernel<128>), grid, dim3(256), lds, s, a); break;
        case 64: hipLaunchKernelGGL((greedy_persist_kernel<64>), grid, dim3(256), lds, s, a); break;
        default:
            set_error("one-launch decode step: head dimension %d not built", hd);
            return TAL_EINVAL;
    }
    TAL_CHECK_LAUNCH("one-launch decode step");
    return TAL_OK;
}

}  // namespace tal

#ifdef PS_TIMELINE
extern "C" int tal_debug_ps_timeline(void* host_out) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(tal::g_ps_tl), sizeof(tal::g_ps_tl)) == hipSuccess ? 0 : -3;
}
#endif
