#!/usr/bin/env python
"""CPU numerics study for a possible round-2 kernel change: 2-term fp16 split (3 MFMAs per product, hi*hi + hi*lo + lo*hi)
with power-of-two scaling per (row, 32-wide k block), against the shipped 3-term bf16 split (6 MFMAs) and plain fp32.
Products are accumulated in fp32 per 32-wide block as the MFMA would, then rescaled and summed in fp32.  Errors vs fp64."""
import numpy as np


def split_bf16_3(x):
    def bf(v):
        u = v.astype(np.float32).view(np.uint32)
        u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
        return u.view(np.float32)
    h = bf(x); r = x - h; m = bf(r); l = bf(r - m)
    return h, m, l


def mm32(a, b):
    return (a.astype(np.float32) @ b.astype(np.float32)).astype(np.float32)


def gemm_bf16x3(A, B, kb=32):
    ah, am, al = split_bf16_3(A); bh, bm, bl = split_bf16_3(B)
    C = np.zeros((A.shape[0], B.shape[1]), np.float32)
    for k0 in range(0, A.shape[1], kb):
        s = slice(k0, k0 + kb)
        for x, y in ((am, bm), (ah, bl), (al, bh), (ah, bm), (am, bh), (ah, bh)):
            C += mm32(x[:, s], y[s])
    return C


def pow2_scale(absmax):
    e = np.floor(np.log2(np.maximum(absmax, 1e-38)))
    return np.exp2(14.0 - e).astype(np.float32)          # block max lands in [2^14, 2^15): inside fp16 range


def gemm_fp16x2(A, B, kb=32):
    C = np.zeros((A.shape[0], B.shape[1]), np.float32)
    for k0 in range(0, A.shape[1], kb):
        a, b = A[:, k0:k0 + kb], B[k0:k0 + kb]
        sa = pow2_scale(np.abs(a).max(1, keepdims=True)); sb = pow2_scale(np.abs(b).max(0, keepdims=True))
        a2, b2 = a * sa, b * sb
        ah = a2.astype(np.float16).astype(np.float32); al = (a2 - ah).astype(np.float16).astype(np.float32)
        bh = b2.astype(np.float16).astype(np.float32); bl = (b2 - bh).astype(np.float16).astype(np.float32)
        acc = mm32(ah, bl) + mm32(al, bh) + mm32(ah, bh)
        C += acc / sa / sb
    return C


def rel(x, ref):
    return float(np.linalg.norm(x - ref) / np.linalg.norm(ref))


def main():
    rng = np.random.RandomState(0)
    M, K, N = 2048, 128, 128
    cases = {
        "N(0,1) x N(0,1/K)": (rng.randn(M, K), rng.randn(K, N) / K ** 0.5),
        "rows spanning 1e-6..1 (gradients)": (rng.randn(M, K) * np.exp(rng.uniform(np.log(1e-6), 0, (M, 1))), rng.randn(K, N) / K ** 0.5),
        "heavy-tailed entries (lognormal)": (rng.randn(M, K) * np.exp(2 * rng.randn(M, K)), rng.randn(K, N) / K ** 0.5),
        "eigenbasis-like (1e-2) x spectrum (1e2)": (rng.randn(M, K) * 1e-2, rng.randn(K, N) * 1e2),
    }
    print("%-44s %12s %12s %12s" % ("case", "fp32", "bf16 x3 (6)", "fp16 x2 (3)"))
    for name, (A, B) in cases.items():
        A, B = A.astype(np.float32), B.astype(np.float32)
        ref = A.astype(np.float64) @ B.astype(np.float64)
        print("%-44s %12.3g %12.3g %12.3g" % (name, rel(mm32(A, B), ref), rel(gemm_bf16x3(A, B), ref), rel(gemm_fp16x2(A, B), ref)))


if __name__ == "__main__":
    main()
