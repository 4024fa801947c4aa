"""Decode control flow on top of the HIP model: the counterpart of `System.generate` and
`System.generate_unaligned` (tal/asr/system.py:68-524).

Same arguments, same return values, same decisions -- restated so that the per-step
device work is what the kernels are good at:
  * the LM / speaker heads run for the last position only (all the loops read,
    system.py:124,355-361);
  * the cross-attention K / V^T of the encoder window are projected once per window
    instead of once per generated token (decoder.cross_kv);
  * log_softmax, the beam top-k and the arg-max run as HIP kernels; the host sees one small
    D2H copy per step (token ids / scores, and for the sliding-window decode the averaged
    attention row it steers by), exactly where the reference synchronises too
    (.cpu()/.tolist()/.item() at system.py:198-215,408-411).
Text-prefix recomputation is kept: the reference decodes with causal_mask=False
(system.py:113,350), so every earlier position attends to later tokens and a self-attention
KV cache would change results (hard parts).

LM fusion (`self.lm`, system.py:127-138) takes any caller-side module called as lm(tokens, causal_mask=False): the
reference's own `tal/lm` package does not exist; pinned with a stand-in LM on both sides (tests/golden/_lm_standin.py).
Not built (not on the acoustic hot path): training / validation steps, data loaders.
The half-precision casts of the waveform (`force_half`, system.py:91-92,285) are performed as written: the model
takes the fp16 waveform, widens it exactly in the front-end and computes in fp32 from there (BASELINE.json: logits
within 1e-3 of the fp32 CPU path run on the same -- fp16-rounded -- audio).
"""
import ctypes as C
import threading
from types import SimpleNamespace

import numpy as np
import torch

from . import _native as N
from . import ops
from .decoder import asr_decode, asr_decode_spk, log_softmax
from .util import ngram_repeat_mask


def _beam_topk(logprobs, scores, done, B, cur_beam, k):
    lib = N.lib()
    V = logprobs.shape[-1]
    dev = logprobs.device
    vals = torch.empty(B, k, dtype=torch.float32, device=dev)
    idx = torch.empty(B, k, dtype=torch.int64, device=dev)
    N.check(lib.tal_beam_topk(N.ptr(logprobs), N.ptr(scores), N.ptr(done), B, cur_beam, V, k, N.ptr(vals),
                              N.ptr(idx), N.stream_handle()), "tal_beam_topk")
    return vals, idx


class _GreedySession:
    """Device-side state of the sliding-window greedy loop for tal_greedy_step_fwd: the decoder's layer structs, the
    embedding / LM-head tensors, one workspace sized for the longest prefix, the {token, attention row} result buffer
    and its pinned host mirror.  `set_window` points it at the cached cross-attention K / V^T of an encoder window."""

    def __init__(self, model, gen_dev, max_positions, sync_mode=2, fold=True):
        from . import decoder as D
        lib = N.lib()
        self.lib = lib
        self.model = model
        stack = model.decoder
        layer0 = stack.layers[0]
        emb = model.embedding.weight
        self.ctx = c = N.GreedyCtx()
        self._arr = D._stack_structs(stack)
        c.layers = C.cast(self._arr, C.c_void_p)
        c.n_layers = len(stack.layers)
        c.E = layer0.linear1.in_features
        c.H = layer0.nhead
        c.FF = layer0.linear1.out_features
        c.V = emb.shape[0]
        c.E0 = model.embed_size or 0
        c.max_len = min(int(max_positions), model.pos_dec_encoder.pe.shape[0])
        c.emb = emb.data_ptr()
        if model.embed_size:
            self._proj_t = D._proj_t(model)
            c.proj, c.proj_t = model.embedding_proj.weight.data_ptr(), self._proj_t.data_ptr()
        c.pe = model.pos_dec_encoder.pe.data_ptr()
        c.no_fold = 0 if fold else 1      # (one form of the decoder layer for the session's lifetime: include/tal_asrd.h, tal_greedy_ctx.no_fold)
        self.dev = emb.device
        self._stream = N.stream_handle()          # the stream current at construction carries every step
        self._tickets = torch.zeros(256, dtype=torch.int32, device=emb.device)   # arrival tickets, self-resetting
        c.tickets = self._tickets.data_ptr()
        self.S = -1
        self.ws = None
        self._ctx_ref = C.byref(self.ctx)
        self._sync_mode = sync_mode      # 2: result written straight to pinned memory and polled; 1: copy command + stream wait
        # may the library loop over this session's steps itself (sync 3: the merged LM-head + pick kernel writes the result word)?
        K0 = c.E0 if c.E0 > 0 else c.E
        self.host_direct_ok = c.E % 16 == 0 and K0 % 8 == 0 and c.emb % 16 == 0 and (not c.proj_t or c.proj_t % 16 == 0)
        self.set_tokens(gen_dev)

    def set_tokens(self, gen_dev):
        self.gen_dev = gen_dev
        self.ctx.tokens = gen_dev.data_ptr()

    def set_window(self, window):
        from . import decoder as D
        mem, mask = window["encoder_out"], window["encoder_padding_mask"]
        D._check_memory(mem, 1, self.ctx.E, "generate_unaligned")
        # K / V^T of the window for every layer, owned by this session (several sessions share one set of weights)
        self._kv = D.stack_kv_private(self.model.decoder, mem)
        self.ctx.k_cache = C.cast(self._kv[0], C.c_void_p)
        self.ctx.vt_cache = C.cast(self._kv[1], C.c_void_p)
        self._kpm = D._kpm_u8(mask, 1, mem.shape[1], self.dev)
        self.ctx.mem_kpm = None if self._kpm is None else self._kpm.data_ptr()
        self.ctx.k_pitch = 0
        self.ctx.kv_all = None
        self._size_for(mem.shape[1])

    def set_episode(self, kv_all, kpm_all, S):
        """Windows as VIEWS of an episode-wide K | V table (include/tal_asrd.h, tal_greedy_ctx.kv_all): kv_all = one [T', 2E]
        tensor per decoder layer (K | V of every encoder frame), kpm_all = uint8 [T'] key-padding bytes or None, S = window
        length.  Afterwards `set_window_frame(frame0)` moves the window with one small launch and no allocation."""
        c = self.ctx
        n = c.n_layers
        self._episode = (kv_all, kpm_all)
        self._kv_arr = (C.c_void_p * n)(*[t.data_ptr() for t in kv_all])
        self._vt = [torch.empty(c.E, self.lib.tal_pad4(S), dtype=torch.float32, device=self.dev) for _ in range(n)]
        self._k_arr = (C.c_void_p * n)()
        self._vt_arr = (C.c_void_p * n)(*[t.data_ptr() for t in self._vt])
        c.kv_all = C.cast(self._kv_arr, C.c_void_p)
        c.kpm_all = None if kpm_all is None else kpm_all.data_ptr()
        c.enc_frames, c.kv_pitch, c.k_pitch = kv_all[0].shape[0], kv_all[0].shape[1], kv_all[0].shape[1]
        c.k_cache = C.cast(self._k_arr, C.c_void_p)
        c.vt_cache = C.cast(self._vt_arr, C.c_void_p)
        self._size_for(S)

    def set_window_frame(self, frame0):
        N.check(self.lib.tal_greedy_set_window(self._ctx_ref, int(frame0), self._stream), "tal_greedy_set_window")

    def _size_for(self, S):
        if S != self.S:
            c = self.ctx
            self.S = c.S = S
            nws = self.lib.tal_greedy_step_workspace_bytes(c.max_len, S, c.E, c.H, c.FF, c.V, c.E0, c.n_layers)
            self.ws = ops._ws(nws, self.dev)
            c.workspace, c.workspace_bytes = self.ws.data_ptr(), nws
            self.picked_dev = torch.empty(1 + S, dtype=torch.float32, device=self.dev)
            self.picked_host = torch.zeros(2 + S, dtype=torch.float32).pin_memory()    # {token, row [S], sequence word}
            self.picked_np = self.picked_host.numpy()[:1 + S]
            c.picked_dev, c.picked_host = self.picked_dev.data_ptr(), self.picked_host.data_ptr()
            c.picked_host_dev = None         # (the library resolves the new buffer's device alias on the next step)

    def step(self, history_start, n_gen):
        """-> (token, attention row [S] float32 copy); the token is also appended at gen_dev[n_gen] on the device."""
        rc = self.lib.tal_greedy_step_fwd(self._ctx_ref, history_start, n_gen, self._sync_mode, self._stream)
        if rc:
            N.check(rc, "tal_greedy_step_fwd")
        return self.result()

    # the same step in two halves, for a host thread that keeps several sessions (one stream each) in flight
    def enqueue(self, history_start, n_gen):
        rc = self.lib.tal_greedy_step_fwd(self._ctx_ref, history_start, n_gen, 3, self._stream)
        if rc:
            N.check(rc, "tal_greedy_step_fwd")

    def ready(self, wait_ms=0):
        got = self.lib.tal_greedy_step_poll(self._ctx_ref, wait_ms)
        if got < 0:
            N.check(got, "tal_greedy_step_poll")
        return got == 1

    def result(self):
        return int(self.picked_np[:1].view(np.int32)[0]), self.picked_np[1:].copy()


_TABLE_LOCK = threading.Lock()      # serialises "does the episode-wide K | V table fit?" + its allocation across runs


class _UnalignedRun:
    """One episode's sliding-window greedy decode (System.generate_unaligned, tal/asr/system.py:254-524) as a state machine, so that
    the same decisions drive a session decoded alone and a session decoded in a group whose steps share their launches
    (System.transcribe_unaligned_many):

        while not run.done:
            run.prepare()                        # window / prefix state on the device for the next step
            token, attn = <one decode step>      # run.step_alone(), or merged steps of several runs (tal_unaligned_group_run)
            run.consume(token, attn)             # the reference's control flow on {token, attention row}

    The control flow itself (system.py:389-521) is ONE implementation, the C host helper tal_unaligned_consume on a plain state
    struct (`self.st`): the loop body runs ~5,700 times per hour of audio beside a ~0.3 ms GPU step, and for a group of sessions
    the library runs step -> poll -> consume without coming back here until a session needs its window moved, its prefix
    uploaded, more room, or is finished.  Token stream and alignment records live in flat numpy buffers the struct points to."""

    def __init__(self, system, audio_x, generated, audio_lens, chunk_size=357, max_iters=1000000, max_positions=None,
                 thresh_prct=0.5, shift_prct=0.25, stall_patience=25, rep_n=5, skip_prct=0.1, fold_layers=True):
        model = system.model
        self.fold_layers = fold_layers
        if generated.size(0) != 1:
            raise ValueError("generate_unaligned handles one episode at a time (system.py:331,411 call .item())")
        self.system, self.model = system, model
        self.dev = dev = audio_x.device
        self.chunk_size = chunk_size
        self.max_positions = model.max_positions if max_positions is None else max_positions
        audio_x = audio_x.half()                        # system.py:285
        with system._lock:
            encoder_out = model.encode(audio_x, audio_lens)
        self.enc, self.mask = encoder_out["encoder_out"], encoder_out["encoder_padding_mask"]
        prime = generated.detach().cpu().numpy().astype(np.int64)[0]
        self.st = st = N.UnalignedState()
        self._st_ref = C.byref(st)
        self._retired = []
        self._gen_t = torch.empty(max(self.HOST_TOKENS0, 2 * prime.size), dtype=torch.int64).pin_memory()    # (pinned: the library uploads
        self.gen = self._gen_t.numpy()                                                                          #  a rewritten prefix itself)
        self.gen[:prime.size] = prime
        cap = self.gen.size
        self.rec_cs = np.empty(cap, dtype=np.int64)               # chunk_start as recorded by the reference
        self.rec_attn = np.empty((cap, chunk_size), dtype=np.float32)   # attention rows (a window never exceeds chunk_size frames)
        self.rec_len = np.empty(cap, dtype=np.int32)
        self._point_buffers()
        st.n = prime.size
        st.encoder_len = int((~self.mask).sum(dim=-1).cpu().item())
        st.eos, st.max_iters = system.tokenizer.eos_token_id, max_iters
        st.chunk_size, st.max_positions, st.stall_patience, st.rep_n = chunk_size, self.max_positions, stall_patience, rep_n
        st.skip_frames, st.shift_frames = int(chunk_size * skip_prct), int(chunk_size * shift_prct)
        st.del_prct = float(np.float32(shift_prct / thresh_prct))
        st.thresh_prct = thresh_prct
        st.flags = N.UNALIGNED_DONE if max_iters <= 0 else 0
        st.gen_pinned = 1
        self.window_key, self.window = None, None
        # K | V of every encoder frame for every decoder layer, once per episode (windows become views: _GreedySession.set_episode)
        # -- whenever whole windows fit the episode; shorter episodes (python-slice windows that wrap) project window by window
        self.kv_all = self.kpm_all = None
        if self.enc.shape[1] >= chunk_size and st.encoder_len >= chunk_size and self.EPISODE_TABLE:
            # the fit decision and the allocations it licenses are ONE step under a lock: runs that start side by side (the worker
            # threads of transcribe_unaligned_many) would otherwise each see the same free memory and each take a quarter of it.
            # The table is optional, so running out of memory while building it is not an error either: the run falls back to
            # window-by-window projection (same results).
            with _TABLE_LOCK:
                if self._table_fits(model):
                    E = self.enc.shape[2]
                    try:
                        kv_all = []
                        for layer in model.decoder.layers:
                            at = layer.multihead_attn
                            bias = torch.cat([at.in_proj_bias.detach()[E:2 * E], torch.zeros(E, dtype=torch.float32, device=dev)])   # (V's bias is added after P.V)
                            kv_all.append(ops.linear(self.enc[0], at.in_proj_weight.detach()[E:3 * E], bias))
                        self.kv_all = kv_all
                        self.kpm_all = self.mask[0].to(torch.uint8).contiguous()
                    except torch.cuda.OutOfMemoryError:
                        kv_all = None
                        self.kv_all = self.kpm_all = None
        self.gen_dev = torch.empty(max(self.DEV_TOKENS0, 2 * prime.size), dtype=torch.int64, device=dev)
        self.dev_len = -1                # how many tokens of the device copy of the stream are current (-1: none)
        self.session, self.session_window = None, None
        self._consume = N.lib().tal_unaligned_consume
        # LM shallow fusion (system.py:368-384): lm_weight * log_softmax(lm(prefix)[-1]) is handed to the step as an additive row
        self.lm_active = system.lm is not None and system.args.lm_weight > 0
        self._lm_bias = None

    HOST_TOKENS0, DEV_TOKENS0 = 4096, 1024      # initial capacities of the token stream (host / device); both grow by doubling
    EPISODE_TABLE = True                        # False: every window is projected on its own (tal_cross_kv_fwd), as in round 3
    TABLE_MAX_FRACTION = 0.25                   # of the memory this process could still get: a table beyond it is not built

    def table_bytes(self, model):
        """Bytes of the episode-wide K | V table: decoder layers x encoder frames x 2E floats (737 MB per hour of audio for 4 x 512)."""
        return len(model.decoder.layers) * self.enc.shape[1] * 2 * self.enc.shape[2] * 4

    def _table_fits(self, model):
        """The table is a speed-up (a window move becomes one launch), not a requirement: with many sessions resident
        (transcribe_unaligned_many keeps streams x group + the queue depth + 1 runs alive) it is only built while it takes at most
        TABLE_MAX_FRACTION of what the device can still give this process (free memory + torch's cached, unallocated blocks);
        otherwise the run projects window by window (12 launches per window move instead of 1, same results)."""
        free, _ = torch.cuda.mem_get_info(self.dev)
        spare = free + torch.cuda.memory_reserved(self.dev) - torch.cuda.memory_allocated(self.dev)
        return self.table_bytes(model) <= self.TABLE_MAX_FRACTION * spare

    def _point_buffers(self):
        st = self.st
        st.gen, st.gen_cap = self.gen.ctypes.data, self.gen.size
        st.rec_chunk_start, st.rec_attn, st.rec_len = self.rec_cs.ctypes.data, self.rec_attn.ctypes.data, self.rec_len.ctypes.data
        st.rec_cap, st.rec_stride = self.rec_cs.size, self.rec_attn.shape[1]

    n = property(lambda self: self.st.n)
    history_start = property(lambda self: self.st.history_start)
    it = property(lambda self: self.st.it)
    done = property(lambda self: bool(self.st.flags & N.UNALIGNED_DONE))

    # -- the device side of the next step
    def prepare(self):
        """Bring the session up to date with what the control flow decided (st.flags): room in the buffers; the encoder window
        [chunk_start, chunk_start + chunk_size) -- python slice semantics as in the reference's slice_tensor, re-materialised
        only when it moves, so that the cross-attention K / V^T cache of every layer keeps hitting --; the device copy of the
        prefix (the kernel that picks a token appends it; the host copy is uploaded again only after a roll-back / forced EOS
        rewrote it)."""
        st = self.st
        n_gen, chunk_start = st.n, st.chunk_start
        assert n_gen - st.history_start <= self.max_positions, "Cannot exceed max context length"
        if st.flags & N.UNALIGNED_PREFIX_REWRITTEN:
            self.dev_len = -1
        elif self.dev_len >= 0:
            self.dev_len = n_gen               # (every token since the last upload was appended on the device by its step)
        if n_gen + 2 >= self.gen.size:         # grow: token stream and records together
            k = self.gen.size
            grown = torch.empty(2 * k, dtype=torch.int64).pin_memory()
            grown.numpy()[:k] = self.gen
            # (the library may still have an upload of a rewritten prefix queued FROM the old buffer, tal_unaligned_group_run: it is
            #  kept until the run ends instead of going back to torch's pinned-memory pool, where another thread could reuse it)
            self._retired.append(self._gen_t)
            self._gen_t, self.gen = grown, grown.numpy()
            self.rec_cs = np.concatenate([self.rec_cs, np.empty(k, dtype=np.int64)])
            self.rec_attn = np.concatenate([self.rec_attn, np.empty((k, self.rec_attn.shape[1]), dtype=np.float32)])
            self.rec_len = np.concatenate([self.rec_len, np.empty(k, dtype=np.int32)])
            self._point_buffers()
        view = self.kv_all is not None and st.it > 0 and 0 <= chunk_start and chunk_start + self.chunk_size <= self.enc.shape[1]
        if self.window_key != chunk_start and not view:
            sl = slice(chunk_start, chunk_start + self.chunk_size)
            self.window = {"encoder_out": self.enc[:, sl].contiguous(), "encoder_padding_mask": self.mask[:, sl].contiguous()}
            self.window_key = chunk_start
        if self.gen_dev.numel() < n_gen + 2:
            self.gen_dev = torch.empty(2 * (n_gen + 2), dtype=torch.int64, device=self.dev)
            self.dev_len = -1
        if self.dev_len != n_gen:
            self.gen_dev[:n_gen] = torch.from_numpy(self.gen[:n_gen])
            self.dev_len = n_gen
        if st.it > 0:
            # every step but the first is ONE C call (tal_greedy_step_fwd, or tal_greedy_step_multi_fwd with other sessions):
            # embed -> decoder stack on the window's cached K / V^T -> LM head of the last position -> pick + append on the
            # device -> {token, attention row} in pinned host memory
            if self.session is None:
                self.session = _GreedySession(self.model, self.gen_dev, self.max_positions, fold=self.fold_layers)
            if self.session.gen_dev is not self.gen_dev:
                self.session.set_tokens(self.gen_dev)
            if view:
                if self.session.ctx.kv_all is None:
                    self.session.set_episode(self.kv_all, self.kpm_all, self.chunk_size)
                    self.session_window = None
                if self.session_window != chunk_start:
                    self.session.set_window_frame(chunk_start)
                    self.session_window = self.window_key = chunk_start
            elif self.session_window is not self.window:
                self.session.set_window(self.window)
                self.session_window = self.window
        st.flags &= N.UNALIGNED_DONE

    def rebind_stream(self):
        """The run continues on the CURRENT stream (it was started on another one and handed over with nothing in flight)."""
        if self.session is not None:
            self.session._stream = N.stream_handle()

    def _lm_logprobs(self, y):
        """system.py:368-384 on the live prefix y [1, U]: the LM never sees speaker tokens (clamped to len(tokenizer) - 1); its
        last-position log-probabilities, times lm_weight -> [1, LM vocabulary].  The LM is the caller's module; the log-softmax
        runs on the HIP row kernel."""
        system = self.system
        lm_input = torch.clamp(y, max=len(system.tokenizer) - 1)
        lm_logits = system.lm(lm_input, causal_mask=False)[:, -1, :]
        return log_softmax(lm_logits.float().contiguous()) * system.args.lm_weight

    def _lm_bias_row(self, y, V):
        """The additive row of tal_greedy_ctx.pick_bias for the step on prefix y: the weighted LM log-probabilities on the shared
        part of the two vocabularies, 0 beyond it."""
        lp = self._lm_logprobs(y)
        if self._lm_bias is None:
            self._lm_bias = torch.zeros(V, dtype=torch.float32, device=self.dev)
        nl = min(lp.size(-1), V)
        self._lm_bias[:nl] = lp[0, :nl]
        return self._lm_bias

    def can_merge(self):
        """May the next step run inside a merged launch (tal_greedy_group_ok)?  Never the first step (module API); never with an
        LM (its forward pass runs between the steps, on this side of the C ABI)."""
        return self.st.it > 0 and not self.lm_active and bool(self.session.lib.tal_greedy_group_ok(self.session._ctx_ref, self.st.history_start, self.st.n))

    def step_alone(self):
        """The step on this session's own launches -> (token, attention row)."""
        st = self.st
        if st.it > 0:
            if self.lm_active:
                # (the device copy of the prefix is current: prepare() uploaded it, the previous step appended its token)
                bias = self._lm_bias_row(self.gen_dev[st.history_start:st.n].view(1, -1), self.session.ctx.V)
                self.session.ctx.pick_bias = bias.data_ptr()
            return self.session.step(st.history_start, st.n)
        # first step through the module API: validates the priming tokens (nn.Embedding raises on out-of-range
        # ids) and the logits (system.py:363-364)
        model, dev, n_gen = self.model, self.dev, st.n
        y = self.gen_dev[st.history_start:n_gen].view(1, -1)
        with self.system._lock:     # (the module API leaves its attention weights on the decoder module)
            logits = asr_decode(model, y, self.window, causal=False, last_only=True, check_tokens=True)  # [1, V]
            all_w = model.decoder.src_attn_weights_all                              # [n_layers, B, U, S]
        if bool(torch.isnan(logits).any()):
            raise Exception("Logits contain nans!")
        if self.lm_active:          # system.py:366-384 as written: log-probabilities, the LM's added on the shared vocabulary
            logits = log_softmax(logits)
            lp = self._lm_logprobs(y)
            nl = min(lp.size(-1), logits.size(-1))
            logits[:, :nl] += lp[:, :nl]
        # token = argmax(log_softmax(logits)) and the attention of the new token averaged over layers (heads
        # are already averaged by the softmax kernel): one launch, one D2H copy (system.py:366-399 does the
        # same arithmetic with a log_softmax, an argmax, a .cpu() per quantity and a numpy mean)
        S_w = all_w.shape[-1]
        picked = torch.empty(1 + S_w, dtype=torch.float32, device=dev)
        N.check(N.lib().tal_greedy_pick_fwd(N.ptr(logits), logits.shape[-1], N.ptr(all_w[0, 0, -1]), all_w.shape[0],
                                            all_w.stride(0), S_w, N.ptr(picked), N.ptr(self.gen_dev[n_gen:]),
                                            N.stream_handle()), "tal_greedy_pick_fwd")
        picked = picked.cpu().numpy()
        return int(picked[:1].view(np.int32)[0]), np.ascontiguousarray(picked[1:], dtype=np.float32)

    def consume(self, token, attn):
        """The reference's control flow on the step's result (system.py:389-521): tal_unaligned_consume."""
        rc = self._consume(self._st_ref, int(token), attn.ctypes.data, attn.shape[0])
        if rc < 0:
            N.check(rc, "tal_unaligned_consume")

    def result(self):
        n, k = self.st.n, self.st.n_rec
        out = torch.from_numpy(self.gen[:n].copy()).view(1, -1).to(self.dev)
        cs_t = torch.from_numpy(self.rec_cs[:k].copy())
        lens = self.rec_len[:k]
        rows = torch.from_numpy(self.rec_attn[:k].copy())
        return out, [(cs_t[i:i + 1], rows[i:i + 1, :int(lens[i])]) for i in range(k)]


class System:
    """Holds the model and the decode-time arguments the reference reads from `self.args`
    (spk_weight, lm_weight) and `self.tokenizer` (eos_token_id)."""

    def __init__(self, model, spk_weight=0.0, eos_token_id=1, bos_token_id=0, pad_token_id=2, tokenizer=None, lm=None, lm_weight=0.0):
        self.model = model
        self.args = SimpleNamespace(spk_weight=spk_weight, lm_weight=lm_weight)
        self.tokenizer = tokenizer if tokenizer is not None else SimpleNamespace(
            eos_token_id=eos_token_id, bos_token_id=bos_token_id, pad_token_id=pad_token_id)
        # shallow fusion in `generate` (system.py:127-138) and `generate_unaligned` (:368-384): any caller-side module called as lm(tokens [rows, U], causal_mask=False)
        # -> logits [rows, U, vocab]; needs a tokenizer with __len__ (speaker tokens are clamped to len(tokenizer) - 1 for the LM)
        self.lm = lm
        # the module API keeps per-call results on the modules themselves (layer.src_attn_weights, the cached window K / V^T):
        # sections that go through it are serialised; the per-token C call of a decode session needs no lock
        self._lock = threading.RLock()

    # transcribe_unaligned_many: sessions that advance in GROUPS (shared launches, group >= 2) decode on the unfolded decoder layer.
    # Same-box A / B on 8 x 1 h (profiles/r6_episode_streams_fold_ab.txt): one session at a time 13.6 -> 12.5 s with the folded layer, four
    # solo sessions in flight 4.56 -> 4.22 s, but 4 threads x groups of 2: 3.27 -> 3.44 s, 2 x 4: 3.52 -> 3.67 s, 32 x 10 min in 4 x 4: 1.77
    # -> 1.94 s -- several merged chains side by side are throughput-bound, and the fold's longer K axis costs there.
    FOLD_GROUP_MAX = 2

    # ------------------------------------------------------------------ several episodes in flight
    @torch.no_grad()
    def transcribe_unaligned_many(self, episodes, streams=None, group=None, stats=None, **kw):
        """`transcribe_unaligned` over a list of episodes with several decode sessions in flight -- the loop the reference runs
        this path in (tal/asr/system.py:625-742 per test item, one after the other).  Each session is the ordinary sliding-window
        decode with its own context (prefix buffer, workspace, window K / V^T, pinned result word); they share the weights.

        group == 1: one host thread, one HIP stream and one chain of launches per session, `streams` of them: the per-token C call
        runs without the interpreter lock, so the launch work of the sessions overlaps, and a decode step (a chain of ~35 small
        dependent kernels on a few dozen CUs) of one session runs beside the others' on the chip.  The device executes at most
        four kernels at a time (its four hardware queues): 3.0x one session at eight streams, a ceiling of 3.9x
        (scripts/ubench/launch_rate.hip, profiles/r3_ubench_launch_rate.txt).

        group > 1: `streams` host threads, each advancing `group` (<= 16) sessions IN STEP through SHARED launches
        (tal_greedy_step_multi_fwd: one chain of 34 launches -- grouped sessions keep the eight-launch decoder layer -- per generated token of every session of the group, each launch
        running the single-session kernel body per session).  A session whose next step does not take the merged kernels' forms
        (first step, prefix beyond 192 tokens, window of 64 frames or fewer) takes that step on launches of its own, in the same
        stream.  While one thread runs the host-side control flow on its group's results, the other threads' steps keep the GPU busy.

        An episode's waveform is uploaded on its session's stream (pinned host memory: the copy runs under the other sessions' compute).

        stats (group > 1): a dict that receives counters of the group loop (calls of tal_unaligned_group_run, merged steps, steps
        taken alone, flags that came back to Python) -- measurement only.

        streams / group left at None: four host threads (the device runs four chains of launches side by side), each with a group
        of ceil(episodes / 4) sessions, at most 4 -- the best measured split for 8 one-hour episodes (4 x 2: 4.0-4.3x the
        one-at-a-time loop); from 32 episodes on, two threads with groups of 16 (32 ten-minute episodes: 6.2x against 5.5x for
        4 x 4), profiles/r5_episode_streams*.txt.

        Device memory: every session in flight (streams x group, plus the producer's queue of max(2, group) + 1 prepared episodes in
        group mode) holds its encoder output and -- while it fits -- an episode-wide K | V table of decoder layers x encoder frames x
        2E floats (737 MB per hour of audio for the reference's 4 x 512 decoder: ~15 GB for 21 one-hour runs).  A run builds the table
        only while it takes at most a quarter of the memory the process can still get (_UnalignedRun._table_fits) and projects window
        by window otherwise: slower window moves, same results.

        episodes: list of (audio [1, L] float tensor -- host (pinned or not) or device --, audio_lens LongTensor [1]).
        Returns [(utterance dicts, generated, alignments)] in episode order: token streams, window starts and utterances identical to
        the solo runs in either mode; attention rows bit-identical to them with group == 1; sessions that advance in groups (group >= 2)
        decode on the unfolded decoder layer: their rows equal a solo run with fold_layers=False bit for bit and the default solo run
        to ~1e-6 (the folded layer re-associates two weight products)."""
        if not episodes:
            return []
        if self.lm is not None and self.args.lm_weight > 0:
            # the LM's forward pass runs between the steps in Python: sessions cannot share launches (the merged loop stays inside the
            # library); they still overlap as one launch chain per session
            group = 1
        if group is None and streams is None and len(episodes) >= 32:
            streams, group = 2, N.TAL_GROUP_MAX         # (a corpus: two chains of 16-session merged steps, 6.2x against 5.5x for 4 x 4)
        if group is None:
            group = 1 if streams is not None else max(1, min(4, -(-len(episodes) // 4)))
        if streams is None:
            streams = 4 if group > 1 else min(8, len(episodes))
        dev = next(self.model.parameters()).device
        # everything the sessions share is built once, on the caller's stream, before any of them starts
        first_audio = episodes[0][0]
        with self._lock:
            warm = first_audio[:, :min(first_audio.shape[1], 16000 * 40)].to(dev)
            self.model.encode(warm, torch.tensor([warm.shape[1]]))
            from . import decoder as D
            D._stack_structs(self.model.decoder)
            if self.model.embed_size:
                D._proj_t(self.model)
        torch.cuda.synchronize(dev)
        results = [None] * len(episodes)
        errors = []
        nxt = [0]
        take = threading.Lock()
        # one stream per host thread, on DIFFERENT hardware queues as far as the device has them (two chains of launches on streams
        # that share a queue run one after the other: 3.3 s or 5.0 s for the same corpus, by the luck of the draw, before hwqueues)
        from . import hwqueues
        group = max(1, min(int(group), N.TAL_GROUP_MAX))
        n_workers = max(1, min(int(streams), (len(episodes) + group - 1) // group)) if group > 1 else 0
        # (group mode: its worker threads + the producer; otherwise one thread per session in flight)
        stream_pool = hwqueues.spread(dev, n_workers + 1 if group > 1 else max(1, min(int(streams), len(episodes))))
        leased = list(stream_pool)
        pool_lock = threading.Lock()

        def own_stream():
            with pool_lock:
                return stream_pool.pop(0) if stream_pool else torch.cuda.Stream(device=dev)

        def worker():
            stream = own_stream()
            with torch.cuda.device(dev), torch.cuda.stream(stream):
                while True:
                    with take:
                        i = nxt[0]
                        nxt[0] += 1
                    if i >= len(episodes) or errors:
                        return
                    try:
                        audio, lens = episodes[i]
                        x = audio.to(dev, non_blocking=True)
                        results[i] = self.transcribe_unaligned(x, lens, **kw)
                        stream.synchronize()
                    except BaseException as e:      # noqa: B902 -- reported to the caller below
                        errors.append((i, e))
                        return

        lib = N.lib()

        import queue
        ready = queue.Queue(maxsize=max(2, group))          # episodes whose session is past its first step

        def producer():
            """Everything an episode needs before it can join a group -- upload, encode, the episode-wide K | V table, the first
            step through the module API, its session's buffers -- on a stream of its own, while the groups keep stepping."""
            stream = producer_stream
            i = -1
            try:
                with torch.cuda.device(dev), torch.cuda.stream(stream):
                    for i, (audio, lens) in enumerate(episodes):
                        if errors:
                            break
                        x = audio.to(dev, non_blocking=True)
                        prime = torch.full((1, 1), self.tokenizer.eos_token_id, dtype=torch.int64, device=dev)
                        # (sessions that advance in groups keep the eight-launch decoder layer for all their steps: FOLD_GROUP_MAX)
                        run = _UnalignedRun(self, x, prime, lens, 357, **dict(kw, fold_layers=kw.get("fold_layers", True) and group < self.FOLD_GROUP_MAX))
                        if not run.done:
                            run.prepare()
                            run.consume(*run.step_alone())
                        stream.synchronize()               # (handed to another stream: nothing of the start-up may be in flight)
                        ready.put((i, run))
            except BaseException as e:      # noqa: B902 -- reported to the caller below
                errors.append((i, e))
            finally:
                for _ in range(n_workers):
                    ready.put(None)

        def group_worker():
            stream = own_stream()
            slots = []                       # [episode index, _UnalignedRun]
            G16 = N.TAL_GROUP_MAX
            st_arr, ctx_arr, cap_arr = (C.POINTER(N.UnalignedState) * G16)(), (C.POINTER(N.GreedyCtx) * G16)(), (C.c_int64 * G16)()
            i, drained = -1, False
            failing = [None]                 # episode index the failing call was made for (None: not attributable to one episode)
            with torch.cuda.device(dev), torch.cuda.stream(stream):
                handle = N.stream_handle()
                try:
                    while True:
                        while len(slots) < group and not drained and not errors:      # refill the free slots with episodes that are ready
                            try:
                                item = ready.get(block=not slots)                     # (wait only when there is nothing to step)
                            except queue.Empty:
                                break
                            if item is None:
                                drained = True
                                break
                            i = item[0]
                            item[1].rebind_stream()
                            slots.append(list(item))
                        for sl in slots:
                            if sl[1].done:
                                results[sl[0]] = self._episode_utterances(*sl[1].result())
                        slots = [sl for sl in slots if not sl[1].done]
                        if errors or (not slots and drained):
                            break
                        if not slots:
                            continue
                        runs = [sl[1] for sl in slots]
                        merged, alone = [], []
                        for sl in slots:
                            r = sl[1]
                            failing[0] = sl[0]
                            if r.st.flags or r.session is None:
                                r.prepare()               # (whatever the control flow asked for: window, prefix upload, room)
                            (merged if r.can_merge() else alone).append(r)
                        if len(merged) < 2:
                            alone, merged = alone + merged, []
                        for r in alone:                   # steps the merged launches do not take
                            failing[0] = next(sl[0] for sl in slots if sl[1] is r)
                            r.prepare()                   # (room for the appended token: nothing else checks it on this path)
                            r.consume(*r.step_alone())
                        failing[0] = None
                        if merged:
                            for k, r in enumerate(merged):
                                st_arr[k], ctx_arr[k], cap_arr[k] = C.pointer(r.st), C.pointer(r.session.ctx), r.gen_dev.numel()
                            # (a free slot: come back soon to look for a ready episode)
                            limit = 4 if alone else (16 if len(slots) < group and not drained else 64)
                            rc = lib.tal_unaligned_group_run(st_arr, ctx_arr, cap_arr, len(merged), limit, handle)
                            if rc < 0:
                                # (a merged call: the library's message names the session; the episodes of the group are reported)
                                failing[0] = [sl[0] for sl in slots if sl[1] in merged]
                                N.check(rc, "tal_unaligned_group_run")
                            if stats is not None:
                                stats["calls"] = stats.get("calls", 0) + 1
                                stats["steps"] = stats.get("steps", 0) + rc
                                stats["sessions"] = stats.get("sessions", 0) + rc * len(merged)
                                stats["alone_steps"] = stats.get("alone_steps", 0) + len(alone)
                                for r in merged:
                                    if r.st.flags:
                                        stats["flag_%d" % r.st.flags] = stats.get("flag_%d" % r.st.flags, 0) + 1
                    stream.synchronize()
                except BaseException as e:      # noqa: B902 -- reported to the caller below
                    errors.append((failing[0] if failing[0] is not None else [sl[0] for sl in slots] or i, e))
                    try:
                        stream.synchronize()    # merged launches may still be in flight: the sessions' workspaces and pinned result
                    except BaseException:       # noqa: B902 -- buffers are released when `slots` goes out of scope
                        pass
                finally:
                    while not drained:          # (after an error: let the producer finish -- it may be blocked on a full queue)
                        drained = ready.get() is None

        if group > 1:
            producer_stream = stream_pool.pop() if len(stream_pool) > n_workers else torch.cuda.Stream(device=dev)    # (the last one dealt: it shares a queue only when the workers already cover them all)
            threads = [threading.Thread(target=producer, daemon=True)] + [threading.Thread(target=group_worker, daemon=True) for _ in range(n_workers)]
        else:
            threads = [threading.Thread(target=worker, daemon=True) for _ in range(max(1, min(int(streams), len(episodes))))]
        try:
            for t in threads:
                t.start()
            for t in threads:
                t.join()
        finally:
            hwqueues.release(leased)
        if errors:
            i, e = errors[0]
            which = "episode %d" % i if isinstance(i, int) else "one of the episodes %s (decoded in one group)" % (list(i),)
            raise RuntimeError("transcribe_unaligned_many: %s failed: %r" % (which, e)) from e
        return results

    # ------------------------------------------------------------------ episode -> utterance dicts
    @torch.no_grad()
    def transcribe_unaligned(self, audio_x, audio_lens, prime=None, **kw):
        """The unaligned branch of `System.test_step` (tal/asr/system.py:654-707) for one episode: sliding-window
        decode primed with EOS ("First token is always EOS"), `tokenizer.decode_speakers` on the generated stream
        without its last token, and per utterance the attention rows / window starts / tokens of
        `alignments[last_split : split + 1]`.  Needs a tokenizer with `decode_speakers` (tal_asrd_amd.tokenizer).
        -> (utterance dicts as pickled by the reference into out/test_result.pkl, generated, alignments)."""
        if prime is None:
            prime = torch.full((1, 1), self.tokenizer.eos_token_id, dtype=torch.int64, device=audio_x.device)
        generated, alignments = self.generate_unaligned(audio_x, prime, audio_lens, chunk_size=357, **kw)
        return self._episode_utterances(generated, alignments)

    def _episode_utterances(self, generated, alignments):
        """(generated, alignments) of generate_unaligned -> (utterance dicts, generated, alignments), system.py:686-707."""
        hyp = generated[0]
        if hyp is None or len(hyp) <= 1:
            return [], generated, alignments
        hyp = hyp[:-1].tolist()
        utts, split_indices = self.tokenizer.decode_speakers(hyp)
        utts = [{"utterance": text, "speakerId": sid} for text, sid in utts]
        last_split_i = 0
        for utt, split_i in zip(utts, split_indices):
            assert split_i - last_split_i > 0
            relevant = alignments[last_split_i:split_i + 1]
            assert len(relevant) > 0
            chunk_starts, weights = zip(*relevant)
            utt["attention"] = torch.cat(weights, dim=0)
            utt["chunkStart"] = torch.cat(chunk_starts, dim=0)
            utt["utteranceTokens"] = hyp[last_split_i:split_i + 1]
            last_split_i = split_i
        return utts, generated, alignments

    # ------------------------------------------------------------------ aligned: batched beam search
    @torch.no_grad()
    def generate(self, audio_x, generated, audio_lens, length, beam_size=1, terminate_token=None,
                 force_half=True, force_output=False):
        """system.py:68-252.  Returns (output_seq, output_spk): per batch item the best
        length-normalised finished sequence (CPU LongTensor | None) and its per-step speaker
        logits (CPU tensor [steps, num_speakers] | None)."""
        model = self.model
        use_spk = self.args.spk_weight > 0
        dev = audio_x.device
        if force_half:
            audio_x = audio_x.half()                    # system.py:91-92
        encoder_out = model.encode(audio_x, audio_lens)
        batch_size = generated.size(0)
        cur_beam = 1
        gen = generated.detach().cpu().numpy().astype(np.int64)           # [rows, len] host bookkeeping
        scores = torch.zeros(batch_size, dtype=torch.float32, device=dev)   # one per live row
        finished = [[] for _ in range(batch_size)]
        done = np.zeros(batch_size * beam_size, dtype=bool)
        spk_embeds = None
        for _ in range(length):
            y = torch.from_numpy(gen).to(dev)
            logits = asr_decode(model, y, encoder_out, causal=False, last_only=True)          # [rows, V]
            V = logits.size(-1)
            pred_speaker = asr_decode_spk(model, y, encoder_out, causal=False, last_only=True) if use_spk else None
            logprobs = log_softmax(logits)
            if self.lm is not None and self.args.lm_weight > 0:
                # system.py:127-138: the LM never sees speaker tokens; its last-position log-probabilities are added on the shared
                # part of the two vocabularies (the LM is the caller's module; its log-softmax runs on the HIP row kernel)
                lm_input = torch.clamp(y, max=len(self.tokenizer) - 1)
                lm_logits = self.lm(lm_input, causal_mask=False)[:, -1, :]
                lm_logprobs = log_softmax(lm_logits.float().contiguous())
                nl = min(lm_logprobs.size(-1), logprobs.size(-1))
                logprobs[:, :nl] += lm_logprobs[:, :nl] * self.args.lm_weight
            done_dev = torch.from_numpy(done.astype(np.uint8)).to(dev) if cur_beam == beam_size else None
            top_scores, indices = _beam_topk(logprobs, scores, done_dev, batch_size, cur_beam, beam_size)
            idx_h = indices.cpu().numpy()
            best_tokens = idx_h % V
            best_beams = idx_h // V
            if cur_beam != beam_size:
                assert beam_size % cur_beam == 0
                rep = beam_size // cur_beam
                gen = np.repeat(gen, rep, axis=0)
                # the reference mutates the caller-visible dict the same way (system.py:168-171);
                # `speaker_out` is NOT repeated there either (latent reference bug for beams + speaker head)
                encoder_out["encoder_out"] = encoder_out["encoder_out"].repeat_interleave(rep, dim=0)
                encoder_out["encoder_padding_mask"] = encoder_out["encoder_padding_mask"].repeat_interleave(rep, dim=0)
            # re-thread every beam onto the hypothesis it extends, then append its token
            src_rows = (np.arange(batch_size)[:, None] * beam_size + best_beams).reshape(-1)
            gen = np.concatenate([gen[src_rows], best_tokens.reshape(-1, 1)], axis=1)
            scores = top_scores.reshape(-1)
            if use_spk:
                if spk_embeds is None:
                    spk_embeds = pred_speaker.unsqueeze(1).repeat_interleave(beam_size // cur_beam, dim=0)
                else:
                    sel = spk_embeds.index_select(0, torch.from_numpy(src_rows).to(dev))
                    spk_embeds = torch.cat((sel, pred_speaker.unsqueeze(1)), dim=1)
                assert gen.shape[1] == spk_embeds.size(1) + 1
            if terminate_token is not None:
                scores_h = None
                for index in np.nonzero(best_tokens.reshape(-1) == terminate_token)[0].tolist():
                    if not done[index]:
                        if scores_h is None:
                            scores_h = scores.cpu()
                        finished[index // beam_size].append(
                            (torch.from_numpy(gen[index].copy()),
                             spk_embeds[index].cpu() if spk_embeds is not None else None, scores_h[index]))
                        done[index] = True
            cur_beam = beam_size
            if done.sum() >= batch_size * beam_size:
                break
        if terminate_token is None or force_output:
            scores_h = scores.cpu().view(batch_size, beam_size)
            for b in range(batch_size):
                for j in range(beam_size):
                    row = b * beam_size + j
                    finished[b].append((torch.from_numpy(gen[row].copy()),
                                        spk_embeds[row].cpu() if spk_embeds is not None else None,
                                        scores_h[b, j].item()))
        finished = [[(cand, spk, score / len(cand)) for cand, spk, score in batch] for batch in finished]
        output_seq = [max(batch, key=lambda x: x[-1])[0] if len(batch) > 0 else None for batch in finished]
        output_spk = [max(batch, key=lambda x: x[-1])[1] if len(batch) > 0 else None for batch in finished]
        return output_seq, output_spk

    # ------------------------------------------------------------------ unaligned: sliding window greedy decode
    @torch.no_grad()
    def generate_unaligned(self, audio_x, generated, audio_lens, chunk_size=357, max_iters=1000000,
                           max_positions=None, thresh_prct=0.5, shift_prct=0.25, stall_patience=25, rep_n=5,
                           skip_prct=0.1, fold_layers=True):
        """system.py:254-524 (designed for batch 1: it calls .item() on per-batch tensors).
        Returns (generated [1, N] LongTensor on the input device, alignments: list of
        (chunk_start LongTensor[1], attention [1, S] CPU tensor) per generated token)."""
        run = _UnalignedRun(self, audio_x, generated, audio_lens, chunk_size, max_iters, max_positions, thresh_prct, shift_prct,
                            stall_patience, rep_n, skip_prct, fold_layers)
        lib = N.lib()
        st_arr, ctx_arr, cap_arr = (C.POINTER(N.UnalignedState) * 1)(), (C.POINTER(N.GreedyCtx) * 1)(), (C.c_int64 * 1)()
        handle = N.stream_handle()
        while not run.done:
            run.prepare()
            if run.st.it == 0 or run.lm_active or not run.session.host_direct_ok:
                # the first step goes through the module API; an LM's forward pass runs between the steps on this side of the C ABI
                run.consume(*run.step_alone())
                continue
            # every other step: step -> poll -> consume inside the library (tal_unaligned_group_run with one session = its own launches),
            # back here only when the control flow asks for something Python owns (a window outside the K | V table, more room, the end):
            # ~10 us of interpreter time per generated token less than the loop above, same calls in the same order
            st_arr[0], ctx_arr[0], cap_arr[0] = C.pointer(run.st), C.pointer(run.session.ctx), run.gen_dev.numel()
            rc = lib.tal_unaligned_group_run(st_arr, ctx_arr, cap_arr, 1, 4096, handle)
            if rc < 0:
                N.check(rc, "tal_unaligned_group_run")
        return run.result()
