"""Host-side byte-pair tokenizers for the MARS5 hot path.

Behavioural mirror of the reference tokenizers (kept on the host):
  * text tokenizer  -> reference ``mars5/minbpe/regex.py:23-164`` (``RegexTokenizer``)
  * L0-code tokenizer -> reference ``mars5/minbpe/codebook.py:14-215`` (``CodebookTokenizer``)
  * "minbpe v1" model text -> reference ``mars5/minbpe/base.py:141-170`` / ``codebook.py:174-206``

Written from scratch around a rank table (pair -> merge order); only the observable
behaviour (ids produced, vocab sizes, special-token handling, ``decode_int``) follows the
reference.  These define token *ids*, not arithmetic, so they are not part of the HIP path.
"""
from __future__ import annotations

import io
from typing import Dict, Iterable, List, Sequence, Tuple, Union

import regex as _re

GPT4_SPLIT_PATTERN = (
    r"""'(?i:[sdmt]|ll|ve|re)|[^\r\n\p{L}\p{N}]?+\p{L}+|\p{N}{1,3}|"""
    r""" ?[^\s\p{L}\p{N}]++[\r\n]*|\s*[\r\n]|\s+(?!\S)|\s+"""
)

_Pair = Tuple[int, int]


def _apply_ranked_merges(symbols: Sequence[int], rank: Dict[_Pair, int], first_new_id: int) -> List[int]:
    """Greedy BPE: repeatedly fuse every occurrence (left to right) of the adjacent pair
    with the lowest rank until no ranked pair is left.  Pair of rank r becomes id
    ``first_new_id + r`` (reference base.py:24-41 + regex.py:90-108 behaviour)."""
    seq = list(symbols)
    while len(seq) > 1:
        best_rank, best_pair = None, None
        for pair in zip(seq, seq[1:]):
            r = rank.get(pair)
            if r is not None and (best_rank is None or r < best_rank):
                best_rank, best_pair = r, pair
        if best_pair is None:
            break
        fused, new_id, i, n = [], first_new_id + best_rank, 0, len(seq)
        a, b = best_pair
        while i < n:
            if i + 1 < n and seq[i] == a and seq[i + 1] == b:
                fused.append(new_id)
                i += 2
            else:
                fused.append(seq[i])
                i += 1
        seq = fused
    return seq


def _read_model_text(src: Union[str, bytes, io.BytesIO, io.StringIO]) -> Tuple[str, Dict[str, int], List[_Pair]]:
    """Parse the "minbpe v1" text format: version / pattern / n_special /
    ``<name> <id>`` lines / ``<a> <b>`` merge lines."""
    if isinstance(src, io.BytesIO):
        text = src.getvalue().decode("utf-8")
    elif isinstance(src, io.StringIO):
        text = src.getvalue()
    elif isinstance(src, bytes):
        text = src.decode("utf-8")
    elif isinstance(src, str) and "\n" not in src:
        assert src.endswith(".model"), "tokenizer model path must end in .model"
        with open(src, encoding="utf-8") as fh:
            text = fh.read()
    else:
        text = src
    lines = text.split("\n")
    assert lines[0].strip() == "minbpe v1", f"unknown tokenizer model version {lines[0]!r}"
    pattern = lines[1].strip()
    n_special = int(lines[2].strip())
    specials: Dict[str, int] = {}
    for ln in lines[3:3 + n_special]:
        name, idx = ln.strip().split()
        specials[name] = int(idx)
    merges: List[_Pair] = []
    for ln in lines[3 + n_special:]:
        if not ln.strip():
            continue
        a, b = ln.split()
        merges.append((int(a), int(b)))
    return pattern, specials, merges


def write_model_text(pattern: str, specials: Dict[str, int], merges: Iterable[_Pair]) -> str:
    out = ["minbpe v1", pattern, str(len(specials))]
    out += [f"{k} {v}" for k, v in specials.items()]
    out += [f"{a} {b}" for a, b in merges]
    return "\n".join(out) + "\n"


class _BPEBase:
    n_base: int = 256

    def __init__(self, pattern: str = None):
        self.pattern = GPT4_SPLIT_PATTERN if pattern is None else pattern
        self.compiled_pattern = _re.compile(self.pattern)
        self.merges: Dict[_Pair, int] = {}
        self.special_tokens: Dict[str, int] = {}
        self.inverse_special_tokens: Dict[int, str] = {}
        self._rank: Dict[_Pair, int] = {}
        self.vocab: Dict[int, bytes] = self._make_vocab()

    # -- vocabulary -----------------------------------------------------------------
    def _base_piece(self, idx: int) -> bytes:
        return bytes([idx])

    def _make_vocab(self) -> Dict[int, bytes]:
        table = {i: self._base_piece(i) for i in range(self.n_base)}
        for (a, b), idx in self.merges.items():
            table[idx] = table[a] + table[b]
        for name, idx in self.special_tokens.items():
            table[idx] = name.encode("utf-8")
        return table

    def register_special_tokens(self, special_tokens: Dict[str, int]) -> None:
        self.special_tokens = dict(special_tokens)
        self.inverse_special_tokens = {v: k for k, v in special_tokens.items()}

    def load(self, model_file) -> None:
        pattern, specials, merge_list = _read_model_text(model_file)
        self.pattern = pattern
        self.merges = {pair: self.n_base + r for r, pair in enumerate(merge_list)}
        self._rank = {pair: r for r, pair in enumerate(merge_list)}
        # NB: like the reference loader, inverse_special_tokens is left untouched here.
        self.special_tokens = specials
        self.vocab = self._make_vocab()

    def dumps(self) -> str:
        ordered = sorted(self.merges.items(), key=lambda kv: kv[1])
        return write_model_text(self.pattern, self.special_tokens, [p for p, _ in ordered])

    # -- encode / decode ------------------------------------------------------------
    def _symbols(self, chunk: str) -> List[int]:
        raise NotImplementedError

    def _chunks(self, text: str) -> List[str]:
        raise NotImplementedError

    def _strip_parts(self) -> bool:
        return False

    def encode_ordinary(self, text: str) -> List[int]:
        ids: List[int] = []
        for chunk in self._chunks(text):
            ids.extend(_apply_ranked_merges(self._symbols(chunk), self._rank, self.n_base))
        return ids

    def encode(self, text: str, allowed_special="none_raise") -> List[int]:
        if allowed_special == "all":
            active = self.special_tokens
        elif allowed_special == "none":
            active = {}
        elif allowed_special == "none_raise":
            active = {}
            assert all(tok not in text for tok in self.special_tokens)
        elif isinstance(allowed_special, set):
            active = {k: v for k, v in self.special_tokens.items() if k in allowed_special}
        else:
            raise ValueError(f"allowed_special={allowed_special} not understood")
        if not active:
            return self.encode_ordinary(text)
        splitter = "(" + "|".join(_re.escape(k) for k in active) + ")"
        ids: List[int] = []
        for part in _re.split(splitter, text):
            if self._strip_parts():
                part = part.strip()
                if not part:
                    continue
            if part in active:
                ids.append(active[part])
            else:
                ids.extend(self.encode_ordinary(part))
        return ids

    def decode(self, ids: Iterable[int]) -> str:
        pieces = []
        for idx in ids:
            if idx in self.vocab:
                pieces.append(self.vocab[idx])
            elif idx in self.inverse_special_tokens:
                pieces.append(self.inverse_special_tokens[idx].encode("utf-8"))
            else:
                raise ValueError(f"invalid token id: {idx}")
        return b"".join(pieces).decode("utf-8", errors="replace")


class RegexTokenizer(_BPEBase):
    """Byte-level BPE over regex-split text (reference regex.py:23)."""
    n_base = 256

    def _chunks(self, text: str) -> List[str]:
        return _re.findall(self.compiled_pattern, text)

    def _symbols(self, chunk: str) -> List[int]:
        return list(chunk.encode("utf-8"))


class CodebookTokenizer(_BPEBase):
    """BPE over space-separated Encodec L0 code integers (reference codebook.py:14).
    Base piece of code c is the bytes of ``" %04d" % c`` so decoded strings split back
    into integers (``decode_int``)."""

    def __init__(self, pattern: str = None, codebook_size: int = 1024):
        self.codebook_size = codebook_size
        self.n_base = codebook_size
        super().__init__(pattern)

    def _base_piece(self, idx: int) -> bytes:
        return f" {idx:04d}".encode("utf-8")

    def _chunks(self, text: str) -> List[str]:
        return [text]

    def _symbols(self, chunk: str) -> List[int]:
        return [int(tok) for tok in chunk.split(" ")]

    def _strip_parts(self) -> bool:
        return True

    def decode_int(self, ids: Iterable[int]) -> List[Union[int, str]]:
        """ids -> list of codebook ints (special tokens come back as their names)."""
        text = self.decode(ids)
        for name in self.special_tokens:
            text = text.replace(name, " " + name + " ")
        out: List[Union[int, str]] = []
        for tok in text.strip().split(" "):
            if not tok:
                continue
            out.append(int(tok) if tok[0].isnumeric() else tok)
        return out

    def expansion_table(self) -> List[List[int]]:
        """token id -> list of L0 codes it expands to (specials -> []).  Lets the host do
        the AR->NAR hand-off as a table lookup instead of string decoding."""
        size = max(self.vocab) + 1
        table: List[List[int]] = [[] for _ in range(size)]
        for idx in range(self.codebook_size):
            table[idx] = [idx]
        for (a, b), idx in sorted(self.merges.items(), key=lambda kv: kv[1]):
            table[idx] = table[a] + table[b]
        return table

    def expansion_csr(self):
        """``expansion_table`` as CSR arrays for the device hand-off (``m5_expand_tokens``): (off int32 (V+1,), vals int64,
        longest run)."""
        import torch
        table = self.expansion_table()
        off = [0]
        vals: List[int] = []
        for run in table:
            vals.extend(run)
            off.append(len(vals))
        return torch.tensor(off, dtype=torch.int32, device="cpu"), torch.tensor(vals or [0], dtype=torch.int64, device="cpu"), max((len(r) for r in table), default=1)


def train_merges(seqs: List[List[int]], n_merges: int, first_new_id: int) -> List[_Pair]:
    """Tiny BPE trainer (most frequent adjacent pair first); used only to make synthetic
    tokenizers whose merges actually fire on test inputs."""
    seqs = [list(s) for s in seqs]
    merges: List[_Pair] = []
    for r in range(n_merges):
        counts: Dict[_Pair, int] = {}
        for s in seqs:
            for pair in zip(s, s[1:]):
                counts[pair] = counts.get(pair, 0) + 1
        if not counts:
            break
        best = max(counts, key=lambda p: (counts[p], -p[0], -p[1]))
        if counts[best] < 2:
            break
        merges.append(best)
        rank = {best: 0}
        seqs = [_apply_ranked_merges(s, rank, first_new_id + r) for s in seqs]
    return merges
