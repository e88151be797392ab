"""Thin Python wrappers over the fp32 train-step kernels (csrc/sgemm_f32.hip, train_ops.hip, losses.hip).

torch tensors are only HBM handles here: every arithmetic operation is a call into liblmrl_amd.so.
"""
from __future__ import annotations

from typing import Optional

from .. import _lib


def _L():
    return _lib.lib()


def _sp():
    return _lib.stream_ptr()


def sgemm(a, b, c, m, n, k, *, trans_a=False, trans_b=False, alpha=1.0, beta=0.0, lda=None, ldb=None, ldc=None,
          a_off=0, b_off=0, c_off=0, batch=(1, 1), sa=(0, 0), sb=(0, 0), sc=(0, 0), bias=None):
    """C = alpha*op(A).op(B) + beta*C (+bias); offsets / strides in ELEMENTS; batch = (outer, inner)."""
    es = 4
    _lib.check(_L().lmrl_sgemm(int(trans_a), int(trans_b), m, n, k, float(alpha), a.data_ptr() + a_off * es, lda, sa[0], sa[1],
                               b.data_ptr() + b_off * es, ldb, sb[0], sb[1], float(beta), c.data_ptr() + c_off * es, ldc, sc[0], sc[1],
                               batch[0], batch[1], _lib.ptr(bias), _sp()), "lmrl_sgemm")


def linear_fwd(x, w, b, y, rows, k, n):
    """y[rows][n] = x[rows][k] @ w[k][n] + b   (flax Dense / HF Conv1D kernel layout [in, out])"""
    sgemm(x, w, y, rows, n, k, lda=k, ldb=n, ldc=n, bias=b)


def linear_bwd(x, w, dy, dx, dw, db, rows, k, n, ws, *, accumulate_dw=True, dx_beta=0.0):
    """dx = dy @ w^T ; dw (+)= x^T @ dy ; db (+)= colsum(dy)"""
    if dx is not None:
        sgemm(dy, w, dx, rows, k, n, trans_b=True, lda=n, ldb=n, ldc=k, beta=dx_beta)
    sgemm(x, dy, dw, k, n, rows, trans_a=True, lda=k, ldb=n, ldc=n, beta=1.0 if accumulate_dw else 0.0)
    if db is not None:
        colsum(dy, rows, n, n, db, accumulate_dw, ws)


def colsum(x, rows, cols, ld, out, accumulate, ws):
    _lib.check(_L().lmrl_colsum(x.data_ptr(), rows, cols, ld, out.data_ptr(), int(accumulate), ws.data_ptr(), _sp()), "lmrl_colsum")


def layernorm_fwd(x, g, b, y, mean, rstd, rows, d, eps):
    _lib.check(_L().lmrl_layernorm_fwd(x.data_ptr(), g.data_ptr(), b.data_ptr(), y.data_ptr(), mean.data_ptr(), rstd.data_ptr(), rows, d,
                                       float(eps), _sp()), "lmrl_layernorm_fwd")


def layernorm_bwd(dy, x, g, mean, rstd, dx, dy_xhat, rows, d, accumulate_dx):
    _lib.check(_L().lmrl_layernorm_bwd(dy.data_ptr(), x.data_ptr(), g.data_ptr(), mean.data_ptr(), rstd.data_ptr(), dx.data_ptr(),
                                       _lib.ptr(dy_xhat), rows, d, int(accumulate_dx), _sp()), "lmrl_layernorm_bwd")


def gelu_fwd(x, y):
    _lib.check(_L().lmrl_gelu_fwd(x.data_ptr(), y.data_ptr(), x.numel(), _sp()), "lmrl_gelu_fwd")


def gelu_bwd(dy, x, dx):
    _lib.check(_L().lmrl_gelu_bwd(dy.data_ptr(), x.data_ptr(), dx.data_ptr(), x.numel(), _sp()), "lmrl_gelu_bwd")


def relu_fwd(x, y):
    _lib.check(_L().lmrl_relu_fwd(x.data_ptr(), y.data_ptr(), x.numel(), _sp()), "lmrl_relu_fwd")


def relu_bwd(dy, x, dx):
    _lib.check(_L().lmrl_relu_bwd(dy.data_ptr(), x.data_ptr(), dx.data_ptr(), x.numel(), _sp()), "lmrl_relu_bwd")


def axpby(a, x, b, y, out):
    _lib.check(_L().lmrl_axpby(float(a), x.data_ptr(), float(b), _lib.ptr(y), out.data_ptr(), out.numel(), _sp()), "lmrl_axpby")


def adamw(p, g, m, v, lr, b1, b2, eps, wd, step):
    _lib.check(_L().lmrl_adamw(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), float(lr), float(b1), float(b2),
                               float(eps), float(wd), int(step), _sp()), "lmrl_adamw")


def embed_fwd(wte, wpe, ids, pos, x, rows, d):
    _lib.check(_L().lmrl_embed_fwd(wte.data_ptr(), wpe.data_ptr(), ids.data_ptr(), pos.data_ptr(), x.data_ptr(), rows, d, _sp()), "lmrl_embed_fwd")


def embed_bwd(dx, ids, pos, dwte, dwpe, rows, d):
    _lib.check(_L().lmrl_embed_bwd(dx.data_ptr(), ids.data_ptr(), pos.data_ptr(), dwte.data_ptr(), dwpe.data_ptr(), rows, d, _sp()), "lmrl_embed_bwd")


def softmax_causal_fwd(s, key_mask, p, batch, heads, t):
    _lib.check(_L().lmrl_softmax_causal_fwd(s.data_ptr(), _lib.ptr(key_mask), p.data_ptr(), batch, heads, t, _sp()), "lmrl_softmax_causal_fwd")


def softmax_bwd(p, dp, rows, t):
    _lib.check(_L().lmrl_softmax_bwd(p.data_ptr(), dp.data_ptr(), rows, t, _sp()), "lmrl_softmax_bwd")


def lse_gather(logits, ld, vocab, targets, rows, logprob=None, lse=None, target_logit=None):
    _lib.check(_L().lmrl_lse_gather(logits.data_ptr(), ld, vocab, targets.data_ptr(), _lib.ptr(logprob), _lib.ptr(lse), _lib.ptr(target_logit),
                                    rows, _sp()), "lmrl_lse_gather")


def ce_bwd(logits, ld, vocab, lse, targets, coef_ce, coef_gather, rows):
    _lib.check(_L().lmrl_ce_bwd(logits.data_ptr(), ld, vocab, lse.data_ptr(), targets.data_ptr(), _lib.ptr(coef_ce), _lib.ptr(coef_gather),
                                rows, _sp()), "lmrl_ce_bwd")


def mask_sum(sta, attn, n, out):
    _lib.check(_L().lmrl_mask_sum(_lib.ptr(sta), _lib.ptr(attn), n, out.data_ptr(), _sp()), "lmrl_mask_sum")
