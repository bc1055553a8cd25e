"""The package's CLIP BPE (tokenizer.ClipBPE: what `load_model(..., vocab=, merges=)` uses for the reference's label prompts, ref src/models.py:155-166)
against `transformers.CLIPTokenizer` built from the SAME vocabulary files.  The real CLIP vocabulary is a download neither box can make, so the files
here are synthetic: every byte symbol (plain and `</w>`-terminated), merges learned by a small BPE training over the prompt words, the two specials last
(as in the real file: <|startoftext|> = V - 2, <|endoftext|> = V - 1)."""
import collections
import json

import numpy as np
import pytest

from owl_vit_object_detection_amd.tokenizer import ClipBPE, bytes_to_unicode, label_prompts

LABELS = ["person", "bicycle", "traffic light", "fire hydrant", "hot dog", "potted plant", "tv", "teddy bear", "hair drier", "N/A"]
EXTRA = ["A photo of  a\tcat's toy??", "naïve café 123 don't we'll", "ÅNGSTRÖM   über", "x" * 40, "", "  ", "emoji 😀 ok", "<|endoftext|> inside"]


def _train_merges(words, n_merges):
    b2u = bytes_to_unicode()
    vocab = collections.Counter()
    for w in words:
        sym = [b2u[b] for b in w.encode("utf-8")]
        sym[-1] += "</w>"
        vocab[tuple(sym)] += 1
    merges = []
    for _ in range(n_merges):
        pairs = collections.Counter()
        for sym, c in vocab.items():
            for a, b in zip(sym[:-1], sym[1:]):
                pairs[(a, b)] += c
        if not pairs:
            break
        best = max(sorted(pairs), key=lambda p: pairs[p])
        merges.append(best)
        nv = collections.Counter()
        for sym, c in vocab.items():
            out, i = [], 0
            while i < len(sym):
                if i < len(sym) - 1 and (sym[i], sym[i + 1]) == best:
                    out.append(sym[i] + sym[i + 1]); i += 2
                else:
                    out.append(sym[i]); i += 1
            nv[tuple(out)] += c
        vocab = nv
    return merges


@pytest.fixture(scope="module")
def vocab_files(tmp_path_factory):
    import regex
    d = tmp_path_factory.mktemp("clipvocab")
    labelmap = {i: l for i, l in enumerate(LABELS)}
    words = []
    for p in label_prompts(labelmap) + EXTRA:
        words += regex.findall(r"[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+", p.lower())
    merges = _train_merges(words, 120)
    b2u = bytes_to_unicode()
    toks = list(b2u.values()) + [c + "</w>" for c in b2u.values()] + [a + b for a, b in merges] + ["<|startoftext|>", "<|endoftext|>"]
    vocab = {}
    for t in toks:
        vocab.setdefault(t, len(vocab))
    (d / "vocab.json").write_text(json.dumps(vocab, ensure_ascii=False), encoding="utf-8")
    (d / "merges.txt").write_text("#version: 0.2\n" + "\n".join(f"{a} {b}" for a, b in merges) + "\n", encoding="utf-8")
    return str(d / "vocab.json"), str(d / "merges.txt"), vocab, merges


def test_ids_equal_hf_clip_tokenizer(vocab_files):
    transformers = pytest.importorskip("transformers")
    vf, mf, vocab, merges = vocab_files
    # OwlViT's tokenizer pads with "!" (id 0).  (HF then treats a literal "!" in a prompt as that special token; no label contains one and the prompts here avoid it.)
    hf = transformers.CLIPTokenizer(vocab=vocab, merges=[tuple(m) for m in merges], pad_token="!")
    ours = ClipBPE(vf, mf)
    prompts = label_prompts({i: l for i, l in enumerate(LABELS)}) + EXTRA
    ref = hf(prompts, padding="max_length", max_length=16, truncation=True, return_tensors="np")["input_ids"]
    got = ours(prompts)
    assert got.shape == ref.shape == (len(prompts), 16) and got.dtype == np.int64
    for p, a, b in zip(prompts, got, ref):
        assert np.array_equal(a, b), (p, a.tolist(), b.tolist())
    # the pooled position the text tower uses (HF5: argmax of the ids = the end token, the highest id) is the end token's FIRST occurrence
    assert (got.argmax(1) == (got == ours.eos).argmax(1)).all()


def test_prompts_follow_the_reference_order():
    assert label_prompts({0: "cat", 1: "dog"}) == ["cat", "a photo of cat", "a cat in an environment", "dog", "a photo of dog", "a dog in an environment"]


def test_load_model_rejects_half_a_vocabulary(vocab_files):
    from owl_vit_object_detection_amd.models import load_model
    with pytest.raises(ValueError, match="vocab.*merges"):
        load_model({0: "cat"}, "cpu", arch="tiny", vocab=vocab_files[0])


@pytest.mark.gpu
def test_load_model_with_vocab_files_equals_prompt_ids_path(vocab_files):
    """`load_model(labelmap, device, vocab=, merges=)` (the unchanged call of ref main.py:42 plus the two CLIP files) builds and tokenises the three prompts per
    label itself and gives the query bank the caller-tokenised `prompt_ids=` path gives -- bit for bit."""
    import torch
    from owl_vit_object_detection_amd.models import load_model
    vf, mf, vocab, merges = vocab_files
    labelmap = {i: l for i, l in enumerate(LABELS[:4])}
    a = load_model(labelmap, "cuda", vocab=vf, merges=mf)             # (default arch = what the reference loads: owlvit-base-patch32, text vocabulary 49 408)
    ids = ClipBPE(vf, mf)(label_prompts(labelmap))
    b = load_model(labelmap, "cuda", prompt_ids=ids)
    qa, qb = dict(a.named_parameters())["queries"], dict(b.named_parameters())["queries"]
    assert qa.shape == (1, 12, a.cfg.text_dim) and torch.equal(qa, qb)
    assert float((qa.norm(dim=-1) - 1).abs().max()) < 1e-3          # L2-normalised text_embeds (HF5:958,970)
    c = load_model(labelmap, "cuda")                                 # without prompts: the seeded random unit rows
    assert not torch.equal(qa, dict(c.named_parameters())["queries"])
