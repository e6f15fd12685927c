"""Pins oracle.longest_first_lengths (the rule csrc/pipeline.hip::ce_build_pairs_kernel implements on the device) against
the `tokenizers` package itself: the same BertWordPieceTokenizer + enable_truncation(strategy="longest_first") call the host
mirror (optimized-rag_amd/cross_encoder.py::tokenize_pairs) and sentence-transformers' CrossEncoder.predict make."""
import numpy as np

from oracle import rag_oracle as O


def test_longest_first_rule_matches_the_tokenizers_package(tmp_path):
    from tokenizers import BertWordPieceTokenizer
    vocab = ["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]", "a", "b"]
    (tmp_path / "vocab.txt").write_text("\n".join(vocab) + "\n")
    tok = BertWordPieceTokenizer(str(tmp_path / "vocab.txt"), lowercase=True)
    rng = np.random.default_rng(0)
    cases = [(n1, n2, L) for L in (8, 9, 16, 31, 32) for n1 in (0, 1, 2, 3, 5, 8, 13, 14, 15, 16, 29, 40) for n2 in (0, 1, 4, 6, 7, 13, 14, 15, 30, 64)]
    cases += [(int(rng.integers(0, 600)), int(rng.integers(0, 600)), 512) for _ in range(40)]
    for n1, n2, L in cases:
        tok.enable_truncation(max_length=L, strategy="longest_first")
        enc = tok.encode(" ".join(["a"] * n1), " ".join(["b"] * n2))
        got = (sum(1 for t in enc.tokens if t == "a"), sum(1 for t in enc.tokens if t == "b"))
        assert got == O.longest_first_lengths(n1, n2, L - 3), (n1, n2, L)
        assert len(enc.ids) == sum(got) + 3
