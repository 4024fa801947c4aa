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

Not built (not on the acoustic hot path): LM fusion (`self.lm`, system.py:127-138 -- no
language model ships with the reference), training / validation steps, data loaders.
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

    def __init__(self, model, gen_dev, max_positions, sync_mode=2):
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
        self.dev = emb.device
        self._stream = N.stream_handle()          # the stream current at construction carries every step
        self._tickets = torch.zeros(256, dtype=torch.int32, device=emb.device)   # arrival tickets, self-resetting
        c.tickets = self._tickets.data_ptr()
        self.S = -1
        self.ws = None
        self._ctx_ref = C.byref(self.ctx)
        self._sync_mode = sync_mode      # 2: result written straight to pinned memory and polled; 1: copy command + stream wait
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
        S = mem.shape[1]
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


class System:
    """Holds the model and the decode-time arguments the reference reads from `self.args`
    (spk_weight, lm_weight) and `self.tokenizer` (eos_token_id)."""

    def __init__(self, model, spk_weight=0.0, eos_token_id=1, bos_token_id=0, pad_token_id=2, tokenizer=None):
        self.model = model
        self.args = SimpleNamespace(spk_weight=spk_weight, lm_weight=0.0)
        self.tokenizer = tokenizer if tokenizer is not None else SimpleNamespace(
            eos_token_id=eos_token_id, bos_token_id=bos_token_id, pad_token_id=pad_token_id)
        self.lm = None
        # the module API keeps per-call results on the modules themselves (layer.src_attn_weights, the cached window K / V^T):
        # sections that go through it are serialised; the per-token C call of a decode session needs no lock
        self._lock = threading.RLock()

    # ------------------------------------------------------------------ several episodes in flight
    @torch.no_grad()
    def transcribe_unaligned_many(self, episodes, streams=8, **kw):
        """`transcribe_unaligned` over a list of episodes with up to `streams` decode sessions in flight -- the loop the
        reference runs this path in (tal/asr/system.py:625-742 per test item, one after the other): each session is the
        ordinary sliding-window decode on its own HIP stream with its own context (prefix buffer, workspace, window K / V^T,
        pinned result word); they share the weights.  One host thread per session: the per-token C call runs without the
        interpreter lock, so the launch work of the sessions overlaps, and a decode step (a chain of ~35 small dependent
        kernels on a few dozen CUs) of one session runs beside the others' on the chip.  An episode's waveform is uploaded
        on its session's stream (pinned host memory: the copy runs under the other sessions' compute).

        `streams` = 8: the device executes at most four kernels at a time (its four hardware queues), and which streams share a
        queue is not under the caller's control -- eight streams fill the four queues whatever the assignment, four may land two
        to a queue (2.5x against 3.0x measured; scripts/ubench/launch_rate.hip, profiles/r3_ubench_launch_rate.txt).

        episodes: list of (audio [1, L] float tensor -- host (pinned or not) or device --, audio_lens LongTensor [1]).
        Returns [(utterance dicts, generated, alignments)] in episode order, identical to the solo runs."""
        if not episodes:
            return []
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

        def worker():
            stream = torch.cuda.Stream(device=dev)
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

        threads = [threading.Thread(target=worker, daemon=True) for _ in range(max(1, min(int(streams), len(episodes))))]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        if errors:
            i, e = errors[0]
            raise RuntimeError("transcribe_unaligned_many: episode %d failed: %r" % (i, e)) from e
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
                           skip_prct=0.1):
        """system.py:254-524 (designed for batch 1: it calls .item() on per-batch tensors).
        Returns (generated [1, N] LongTensor on the input device, alignments: list of
        (chunk_start LongTensor[1], attention [1, S] CPU tensor) per generated token)."""
        model = self.model
        if generated.size(0) != 1:
            raise ValueError("generate_unaligned handles one episode at a time (system.py:331,411 call .item())")
        dev = audio_x.device
        max_positions = model.max_positions if max_positions is None else max_positions
        audio_x = audio_x.half()                        # system.py:285
        with self._lock:
            encoder_out = model.encode(audio_x, audio_lens)
        enc, mask = encoder_out["encoder_out"], encoder_out["encoder_padding_mask"]
        encoder_len = int((~mask).sum(dim=-1).cpu().item())
        eos = self.tokenizer.eos_token_id
        # Host bookkeeping in flat buffers (the loop body runs ~5000 times per hour of audio next to a ~0.3 ms GPU step):
        # the token stream in a numpy array of length n, the alignment records as two parallel lists.
        prime = generated.detach().cpu().numpy().astype(np.int64)[0]
        gen = np.empty(max(4096, 2 * prime.size), dtype=np.int64)
        n = prime.size
        gen[:n] = prime
        rec_cs, rec_attn = [], []            # chunk_start as recorded by the reference; attention row
        chunk_start = 0
        history_start = 0
        highest_progress = 0
        num_no_improve = 0
        window_time = 0
        window_key, window = None, None
        gen_dev = torch.empty(max(1024, 2 * n), dtype=torch.int64, device=dev)
        dev_len = -1
        session, session_window = None, None
        ngram_count = N.lib().tal_ngram_repeat_count
        gen_addr = gen.ctypes.data
        attn_range, attn_range_S = None, -1
        for it in range(max_iters):
            hist_len = n - history_start             # the model input of this step (before the new token is appended)
            assert hist_len <= max_positions, "Cannot exceed max context length"
            # encoder window [chunk_start, chunk_start + chunk_size) -- python slice semantics as in the
            # reference's slice_tensor; re-materialised only when the window moves so that the
            # cross-attention K / V^T cache of every layer keeps hitting
            if window_key != chunk_start:
                sl = slice(chunk_start, chunk_start + chunk_size)
                window = {"encoder_out": enc[:, sl].contiguous(), "encoder_padding_mask": mask[:, sl].contiguous()}
                window_key = chunk_start
            # the prefix lives on the device: the kernel that picks a token appends it, and the host copy is
            # uploaded again only after the control flow below rewrote it (roll-back, forced EOS)
            n_gen = n
            if gen_dev.numel() < n_gen + 1:
                gen_dev = torch.empty(2 * (n_gen + 1), dtype=torch.int64, device=dev)
                dev_len = -1
            if gen.size < n_gen + 1:
                gen = np.concatenate([gen, np.empty(gen.size, dtype=np.int64)])
                gen_addr = gen.ctypes.data
            if dev_len != n_gen:
                gen_dev[:n_gen] = torch.from_numpy(gen[:n_gen])
                dev_len = n_gen
            if it == 0:
                # first step through the module API: validates the priming tokens (nn.Embedding raises on out-of-range
                # ids) and the logits (system.py:363-364)
                y = gen_dev[history_start:n_gen].view(1, -1)
                with self._lock:     # (the module API leaves its attention weights on the decoder module)
                    logits = asr_decode(model, y, window, causal=False, last_only=True, check_tokens=True)  # [1, V]
                    all_w = model.decoder.src_attn_weights_all                              # [n_layers, B, U, S]
                if bool(torch.isnan(logits).any()):
                    raise Exception("Logits contain nans!")
                # token = argmax(log_softmax(logits)) and the attention of the new token averaged over layers (heads
                # are already averaged by the softmax kernel): one launch, one D2H copy (system.py:366-399 does the
                # same arithmetic with a log_softmax, an argmax, a .cpu() per quantity and a numpy mean)
                S_w = all_w.shape[-1]
                picked = torch.empty(1 + S_w, dtype=torch.float32, device=dev)
                N.check(N.lib().tal_greedy_pick_fwd(N.ptr(logits), logits.shape[-1], N.ptr(all_w[0, 0, -1]), all_w.shape[0],
                                                    all_w.stride(0), S_w, N.ptr(picked), N.ptr(gen_dev[n_gen:]),
                                                    N.stream_handle()), "tal_greedy_pick_fwd")
                picked = picked.cpu().numpy()
                token = int(picked[:1].view(np.int32)[0])
                attn = picked[1:].astype(np.float32)
            else:
                # every later step is ONE C call (tal_greedy_step_fwd): embed -> decoder stack on the window's cached
                # K / V^T -> LM head of the last position -> pick + append on the device -> {token, attention row}
                # in pinned host memory
                if session is None:
                    session = _GreedySession(model, gen_dev, max_positions)
                if session.gen_dev is not gen_dev:
                    session.set_tokens(gen_dev)
                if session_window is not window:
                    session.set_window(window)
                    session_window = window
                token, attn = session.step(history_start, n_gen)
            gen[n] = token
            n += 1
            dev_len = n
            rec_cs.append(chunk_start)
            rec_attn.append(attn)
            assert len(rec_cs) == n - 1
            S = attn.shape[0]
            if S != attn_range_S:
                attn_range, attn_range_S = (np.arange(S, dtype=np.float32) / np.float32(S)).astype(np.float32), S
            prct_progress = float(np.sum(attn * attn_range, dtype=np.float32))
            if prct_progress > highest_progress:
                num_no_improve = 0
                if window_time > 5:
                    highest_progress = prct_progress
            else:
                num_no_improve += 1
            is_stalling = num_no_improve >= stall_patience
            # ngram_repeat_mask(model_input, rep_n).sum() over the step's input (tal/asr/util.py:5-17, system.py:418-421)
            rep_count = ngram_count(gen_addr + 8 * history_start, hist_len, rep_n)
            is_repeating = rep_count > rep_n * 2
            is_last_chunk = encoder_len - chunk_start <= chunk_size
            reset_window = is_stalling or is_repeating
            record_kept = True
            if not is_last_chunk:
                if reset_window:
                    chunk_start += int(chunk_size * skip_prct)
                    if is_repeating:
                        rollback = 2 * rep_n
                        n -= rollback - 1
                        del rec_cs[-(rollback - 1):]
                        del rec_attn[-(rollback - 1):]
                        record_kept = False            # this step's record is among the ones rolled back
                    gen[n - 1] = eos
                    dev_len = -1                   # the device copy of the prefix is stale
                    history_start = n - 1
                    highest_progress = 0
                    window_time = 0
                elif prct_progress > thresh_prct:
                    history_size = n - history_start
                    chunk_start += int(chunk_size * shift_prct)
                    history_start += int(np.floor(np.float32(shift_prct / thresh_prct) * np.float32(history_size - 1)))
                    highest_progress = 0
                    window_time = 0
            # The reference stores the chunk_start *tensor object* in `alignments` and then advances it in
            # place (system.py:400,441,468), so the recorded value is the post-advance, pre-clamp one.
            if record_kept:
                rec_cs[-1] = chunk_start
            chunk_start = min(chunk_start, encoder_len - chunk_size)
            history_start = max(history_start, max(n - max_positions, 0))
            assert history_start < n, ("Invalid history start index", history_start, n)
            assert n - history_start <= max_positions, ("Exceed max positions", history_start, n)
            window_time += 1
            if reset_window and is_last_chunk:
                break
        out = torch.from_numpy(gen[:n].copy()).view(1, -1).to(dev)
        rows = torch.from_numpy(np.stack(rec_attn)) if len({a.shape[0] for a in rec_attn}) == 1 else None
        cs_t = torch.tensor(rec_cs, dtype=torch.int64)
        return out, [(cs_t[i:i + 1], rows[i:i + 1] if rows is not None else torch.from_numpy(rec_attn[i]).unsqueeze(0))
                     for i in range(len(rec_cs))]
