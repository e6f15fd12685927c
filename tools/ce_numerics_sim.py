"""Diagnostic (not product): CPU simulation of the cross-encoder's operand formats, to choose the MFMA operand scheme before
any kernel is written. Every GEMM-shaped product of the forward (QKV, Q.K^T, P.V, out-proj, FFN-up, FFN-down) is computed in
float64 from operands ROUNDED the way a scheme stores them; everything else is exact float64. The logits are compared with the
exact float64 forward (oracle/bert_oracle.py - test infrastructure; this tool is one of its allowed users: tools/ is not product).

Schemes (x = hi + lo, hi = fp16(x)):
  f16      hi.hi only                                                    (1 MFMA unit;  the r1 experiment: fails the bar)
  split16  lo = fp16(x - hi); hi.hi + lo.hi + hi.lo                      (3 units: what rounds 1-3 ship)
  e4m3     hi.hi in fp16 + lo8.hi8 + hi8.lo8, hi8 = e4m3(hi), lo8 = e4m3(lo * 2^11)   (2 units, 4 B / element)
  bf8t     as e4m3 but hi8 = the TOP BYTE of hi (e5m2 by truncation: free from the fp16 register by v_perm), lo8 scaled by
           the mean truncation loss                                      (2 units, 3 B / element)
  bf8r     hi8 = e5m2(hi) round-to-nearest (top byte of hi + 0x80)       (2 units, 3 B / element, 2 more VALU per dword)
  bf8tt    hi8 = top byte of hi (truncation), lo8 = e5m2(lo * 2^11 * gain): both correction operands in ONE 8-bit format, so the
           two correction products of a K range can share one block-scaled MFMA ([lo8 | hi8] . [hi8 | lo8])   (2 units, 3 B)
  bf8tt_stream  bf8tt, and every stored activation (residual stream, ctx, FFN intermediate) read back as hi16 + lo8
  ship     bf8tt_stream with attention in split fp16 (what the first MX build shipped); ship_noplo: without the P_lo term of P.V
  shiprn   ship with hi8 rounded to NEAREST (top byte of hi + 0x80) and no gain on lo8: what ships now
  shipfp6_<e2m3|e3m2>_<group>  the correction operands as 6-bit floats (twice the MFMA rate of 8-bit ones) under one E8M0 scale per
           <group> consecutive K elements; lo6 = fp6(lo * 2^11 / scale) shares the scale of hi6; stored activations = hi16 + lo6
usage: python tools/ce_numerics_sim.py [pairs] [L] [scheme,scheme...]"""
import math
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import bert_oracle as B  # noqa: E402


def f16(x):
    return x.astype(np.float16).astype(np.float64)


def q8(x, dt):
    lim = 448.0 if dt == torch.float8_e4m3fn else 57344.0
    t = torch.from_numpy(np.clip(x, -lim, lim).astype(np.float32))
    return t.to(dt).to(torch.float32).numpy().astype(np.float64)


def top_byte(hi, rnd):
    """fp16 value -> the e5m2 value its top byte encodes (rnd: add 0x80 to the bit pattern first = round half up in magnitude)"""
    u = hi.astype(np.float16).view(np.uint16).astype(np.uint32)
    if rnd:
        u = u + 0x80
    u = (u & 0xFF00).astype(np.uint16)
    return u.view(np.float16).astype(np.float64)


def q6(v, mb, emin, vmax):
    """round to a 6-bit float grid: mb mantissa bits, smallest normal exponent emin (below it the subnormal step), clipped to vmax"""
    a = np.abs(v)
    e = np.floor(np.log2(np.maximum(a, 2.0 ** emin)))
    step = 2.0 ** (e - mb)
    return np.clip(np.rint(v / step) * step, -vmax, vmax)


FP6 = {"e2m3": (3, 0, 7.5), "e3m2": (2, -2, 28.0)}


def fp6_pair(hi, lo, fmt, group):
    """(hi6, lo6) as real values: both in one 6-bit format under ONE power-of-two scale per `group` consecutive K elements (the
    scale byte a lane hands the block-scaled MFMA), lo carried as lo * 2^11 so it shares the scale of hi"""
    mb, emin, vmax = FP6[fmt]
    sh = hi.shape
    g = hi.reshape(sh[:-1] + (sh[-1] // group, group))
    l = (lo * 2048.0).reshape(g.shape)
    m = np.maximum(np.abs(g).max(-1, keepdims=True), 1e-30)
    sc = 2.0 ** np.ceil(np.log2(m / vmax))
    return (q6(g / sc, mb, emin, vmax) * sc).reshape(sh), (q6(l / sc, mb, emin, vmax) * sc / 2048.0).reshape(sh)


TRUNC_GAIN = 1.0 / 0.915       # mean of hi / trunc(hi) for uniformly distributed low mantissa bits


class Scheme:
    def __init__(self, name):
        self.name = name

    def split(self, x, wscale=1.0):
        """-> (hi, lo_for_correction, hi_for_correction) as float64 arrays holding the ROUNDED values"""
        hi = f16(x)
        lo = x - hi
        n = self.name
        if n == "f16":
            return hi, None, None
        if n == "split16":
            return hi, f16(lo), hi
        if n == "e4m3":
            return hi, q8(lo * 2048.0 * wscale, torch.float8_e4m3fn) / (2048.0 * wscale), q8(hi * wscale, torch.float8_e4m3fn) / wscale
        if n == "bf8t":
            return hi, q8(lo * 2048.0 * wscale * TRUNC_GAIN, torch.float8_e4m3fn) / (2048.0 * wscale), top_byte(hi, False)
        if n == "bf8r":
            return hi, q8(lo * 2048.0 * wscale, torch.float8_e4m3fn) / (2048.0 * wscale), top_byte(hi, True)
        if n in ("bf8tt", "bf8tt_stream", "ship", "ship_noplo"):     # hi8 = top byte of hi, lo8 = e5m2(lo * 2^11 * gain) round-to-nearest: ONE operand format (bf8)
            return hi, q8(lo * 2048.0 * wscale * TRUNC_GAIN, torch.float8_e5m2) / (2048.0 * wscale), top_byte(hi, False)
        if n == "shiprn":                        # ship with hi8 rounded to nearest (top byte of hi + 0x80), no gain
            return hi, q8(lo * 2048.0 * wscale, torch.float8_e5m2) / (2048.0 * wscale), top_byte(hi, True)
        if n.startswith("shipfp6"):             # shipfp6_<fmt>_<group>: hi16 + (hi6, lo6) pairs under a per-group scale, attention in split fp16
            _, fmt, grp = n.split("_")
            h6, l6 = fp6_pair(hi, lo, fmt, int(grp))
            return hi, l6, h6
        if n == "bf8t_nogain":
            return hi, q8(lo * 2048.0 * wscale, torch.float8_e4m3fn) / (2048.0 * wscale), top_byte(hi, False)
        raise ValueError(n)

    def store(self, x):
        """what a consumer reads back from an activation tensor stored as hi16 + lo8 (the residual stream, ctx, the FFN intermediate)"""
        if not (self.name.endswith("_stream") or self.name.startswith("ship")):
            return x
        hi = f16(x)
        if self.name.startswith("shipfp6"):
            _, fmt, grp = self.name.split("_")
            return hi + fp6_pair(hi, x - hi, fmt, int(grp))[1]
        g = 1.0 if self.name == "shiprn" else TRUNC_GAIN
        return hi + q8((x - hi) * 2048.0 * g, torch.float8_e5m2) / (2048.0 * g)

    def mm(self, a, b, a_scale=1.0, b_scale=1.0, drop_a_lo=False):
        """a @ b^T over the last axis of both, operands rounded per scheme (leading axes broadcast as numpy matmul)"""
        ah, al, a8 = self.split(a, a_scale)
        bh, bl, b8 = self.split(b, b_scale)
        bt = lambda t: np.swapaxes(t, -1, -2)
        y = ah @ bt(bh)
        if al is not None:
            y = y + a8 @ bt(bl)
            if not drop_a_lo:
                y = y + al @ bt(b8)
        return y


def wscale_for(w):
    """power-of-two scale that puts max|w| just under e4m3's 448 (weights are ~0.05: unscaled they would sit in e4m3 subnormals)"""
    return 2.0 ** math.floor(math.log2(448.0 / np.abs(w).max()))


def forward(w, cfg, ids, tt, lens, sch, sites=None):
    from scipy.special import erf
    W = {k: v.astype(np.float64) for k, v in w.items()}
    P, L = ids.shape
    H, nh = cfg["hidden"], cfg["heads"]
    dh = H // nh
    x = (W["bert.embeddings.word_embeddings.weight"][ids] + W["bert.embeddings.token_type_embeddings.weight"][tt]
         + W["bert.embeddings.position_embeddings.weight"][np.arange(L)][None])
    x = sch.store(B._ln(x, W["bert.embeddings.LayerNorm.weight"], W["bert.embeddings.LayerNorm.bias"], cfg["eps"]))
    key_ok = np.arange(L)[None, :] < np.asarray(lens)[:, None]
    add_mask = np.where(key_ok, 0.0, -1e30)[:, None, None, :]
    lin = lambda t, name: sch.mm(t, W[name + ".weight"], 1.0, wscale_for(W[name + ".weight"])) + W[name + ".bias"]
    for l in range(cfg["layers"]):
        p = f"bert.encoder.layer.{l}."
        q = lin(x, p + "attention.self.query")
        k = lin(x, p + "attention.self.key")
        v = lin(x, p + "attention.self.value")
        sp = lambda t: t.reshape(P, L, nh, dh).transpose(0, 2, 1, 3)
        s = sch.mm(sp(q), sp(k)) * (dh ** -0.5) + add_mask
        s = s - s.max(-1, keepdims=True)
        e = np.exp(s)                                   # unnormalised P in (0, 1], as the kernel's online softmax holds it
        # ship*: the shipped forward keeps attention in split fp16 whatever the GEMM scheme; *_noplo: P.V without the P_lo term
        att = Scheme("split16") if sch.name.startswith("ship") else sch
        if sch.name.startswith("ship"):
            s = att.mm(sp(q), sp(k)) * (dh ** -0.5) + add_mask
            s = s - s.max(-1, keepdims=True)
            e = np.exp(s)
        ctx = att.mm(e, np.swapaxes(sp(v), -1, -2), drop_a_lo=sch.name.endswith("noplo")) / e.sum(-1, keepdims=True)
        ctx = sch.store(ctx.transpose(0, 2, 1, 3).reshape(P, L, H))
        o = lin(ctx, p + "attention.output.dense")
        x = sch.store(B._ln(o + x, W[p + "attention.output.LayerNorm.weight"], W[p + "attention.output.LayerNorm.bias"], cfg["eps"]))
        h = lin(x, p + "intermediate.dense")
        h = sch.store(0.5 * h * (1.0 + erf(h / math.sqrt(2.0))))
        o = lin(h, p + "output.dense")
        x = sch.store(B._ln(o + x, W[p + "output.LayerNorm.weight"], W[p + "output.LayerNorm.bias"], cfg["eps"]))
    pooled = np.tanh(x[:, 0] @ W["bert.pooler.dense.weight"].T + W["bert.pooler.dense.bias"])
    return (pooled @ W["classifier.weight"].T + W["classifier.bias"])[:, 0]


def main():
    P = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    L = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    cfg = B.minilm_config()
    rng = np.random.default_rng(7)
    lens = (18 + rng.integers(96, 225, P)).clip(max=L).astype(np.int32)
    lens[:8] = [L, L, 5, 17, 64, 200, 33, 128][:min(8, P)]
    print("| scheme | max abs logit err (seed 99 / 2024) | rms |")
    print("|---|---|---|")
    rows = {}
    for seed in (99, 2024):
        w = B.seeded_weights(cfg, seed)
        ids = rng.integers(1000, cfg["vocab_size"], (P, L)).astype(np.int64)
        ids[np.arange(L)[None, :] >= lens[:, None]] = 0
        tt = ((np.arange(L)[None, :] >= 18) & (np.arange(L)[None, :] < lens[:, None])).astype(np.int64)
        exp = B.forward_logits(w, cfg, ids, tt, lens, fast_erf=True)
        names = sys.argv[3].split(",") if len(sys.argv) > 3 else ("f16", "split16", "e4m3", "bf8r", "bf8t", "bf8t_nogain", "bf8tt", "bf8tt_stream")
        for name in names:
            got = forward(w, cfg, ids, tt, lens, Scheme(name))
            err = got - exp
            rows.setdefault(name, []).append((np.abs(err).max(), math.sqrt((err ** 2).mean())))
    for name, r in rows.items():
        print(f"| {name} | {r[0][0]:.2e} / {r[1][0]:.2e} | {r[0][1]:.2e} / {r[1][1]:.2e} |")


if __name__ == "__main__":
    main()
