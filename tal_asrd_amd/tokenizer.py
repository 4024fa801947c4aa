"""Host-side mirror of the reference's tokenizer interface (tal/asr/tokenizers/__init__.py), as far as the
decode path touches it: the special ids, `decode` / `decode_list`, and `decode_speakers`, which turns a
generated token stream into (utterance text, speaker) pairs and the EOS positions (= speaker-change indices)
that `System.test_step` aligns the attention records to (tal/asr/system.py:688-707).

The reference's concrete tokenizer wraps a sentencepiece model (`taltoken-cased.model`,
tal/asr/tokenizers/sentencepiece.py) that does not ship with the repository.  `SynthTokenizer` stands in for
it with the deterministic piece table of tal_asrd_amd.synth (same conventions: a piece starts a word or
continues one); a real sentencepiece model plugs in through `PieceTokenizer(decode_ids=sp.DecodeIds, ...)`.
"""
from . import synth


class Tokenizer:
    """tal/asr/tokenizers/__init__.py:7-138 (decode side)."""

    def __init__(self, bos_token_id=0, eos_token_id=1, pad_token_id=2, eot_token_id=49129):
        self._bos_token_id = bos_token_id
        self._eos_token_id = eos_token_id
        self._pad_token_id = pad_token_id
        self._eot_token_id = eot_token_id

    def __len__(self):
        return 0

    @property
    def pad_token_id(self):
        return self._pad_token_id

    @property
    def bos_token_id(self):
        return self._bos_token_id

    @property
    def eos_token_id(self):
        return self._eos_token_id

    @property
    def eot_token_id(self):
        return self._eot_token_id

    def decode_list(self, tokens):
        raise NotImplementedError

    def decode(self, tokens):
        """:87-101: accepts a list or a tensor."""
        if hasattr(tokens, "cpu"):
            tokens = tokens.cpu().tolist()
        return self.decode_list(list(tokens))

    def decode_speakers(self, tokens, add_last=True):
        """:103-138.  Ids >= len(self) are speaker tokens, EOS closes an utterance, BOS is ignored.
        -> ([(text, speaker | None), ...], [token index of every split]).  The token-level loop is util.split_speaker_turns (one
        implementation for the scorer and the tokenizer); each turn's tokens are then rendered by `decode`."""
        from .util import split_speaker_turns
        turns, split_indices = split_speaker_turns(tokens, len(self), self.bos_token_id, self.eos_token_id, add_last)
        utterances = [(self.decode(buf), speaker) for buf, speaker in turns]
        assert len(utterances) == len(split_indices)
        return utterances, split_indices


class PieceTokenizer(Tokenizer):
    """tal/asr/tokenizers/sentencepiece.py:17-85 over any `decode_ids(list[int]) -> str` (sentencepiece's
    DecodeIds): <EOT> and speaker tokens are rendered inline, everything else goes through decode_ids."""

    def __init__(self, decode_ids, vocab_size, bos_token_id=0, eos_token_id=1, pad_token_id=2):
        super().__init__(bos_token_id, eos_token_id, pad_token_id, eot_token_id=0)   # sentencepiece.py:29
        self._decode_ids = decode_ids
        self._vocab = int(vocab_size)

    def __len__(self):
        return self._vocab

    def decode_list(self, tokens):
        out, buf = "", []
        for x in tokens:
            x = int(x)
            clear = x == self.eot_token_id or x >= len(self)
            if clear:
                if buf:
                    out += self._decode_ids(buf)
                buf = []
            if x == self.eot_token_id:
                out += "<EOT>"
            elif x >= len(self):
                out += "<S{}>".format(x - len(self))
            else:
                buf.append(x)
        if buf:
            out += self._decode_ids(buf)
        return out


class SynthTokenizer(PieceTokenizer):
    def __init__(self, vocab_size=10000):
        super().__init__(synth.decode_pieces, vocab_size)
