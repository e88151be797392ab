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


def _pad(n: int, m: int = 64) -> int:
    return -(-n // m) * m


def _padn(n: int) -> int:
    """Padded N extent (rows of a staged W operand): vocabulary-sized extents go to multiples of 256 so that the 256x256-tile kernel
    applies; everything else to multiples of 64."""
    return _pad(n, 256) if n >= 8192 else _pad(n)


def _pitch(k: int) -> int:
    """Row pitch (elements) of a staged bf16 operand with K extent k: pad64(k), plus 64 elements when that is a multiple of 512 elements
    (1 KiB): with a power-of-two pitch every row of a GEMM tile starts in the same HBM channel (measured 3x on the dW products, K = B*T)."""
    kp = _pad(k)
    return kp + 64 if kp % 512 == 0 else kp


class MatmulBF16:
    """The train step's bf16-MFMA matmul mode (the reference's optional `bf16_activations`, train_ilql_gpt2.py:193): every Dense / Conv1D
    product of forward and backward runs on the rollout engine's bf16 GEMM kernels (`lmrl_gemm_bf16`: fp32 accumulation, fp32 outputs);
    parameters, gradients, optimizer state, LayerNorm / softmax / losses and the attention products stay fp32.  This object owns the
    staged bf16 operands: weight copies (and their transposes) are cast once per step (`begin_step()` drops them: the optimizer has moved
    the fp32 masters), activation / gradient operands go through reusable scratch buffers (csrc/train_bf16.hip)."""

    def __init__(self, device):
        import torch
        self.t, self.dev = torch, device
        self.w, self.scratch = {}, {}

    def begin_step(self):
        self.w.clear()

    def stage_arena(self, flat, mats, transposed: bool):
        """Stage the bf16 copies of ALL the Dense kernels `mats` (fp32 [in][out] views into the parameter arena `flat`) in one launch
        (lmrl_cast_bf16_segments) and register them as this step's kept copies: transposed=True the forward operands ("wT": [pad(out)][pitch(in)]),
        False the dX operands ("w": [pad(in)][pitch(out)]) — what `cast(..., keep=True)` would produce one launch per matrix."""
        import numpy as np
        if not hasattr(self, "_plans"):
            self._plans = {}
        key = (flat.data_ptr(), bool(transposed), tuple(w.data_ptr() for w in mats))
        plan = self._plans.get(key)
        if plan is None:
            dt = np.dtype([("src_off", "<i8"), ("nat_off", "<i8"), ("t_off", "<i8"), ("rows", "<i4"), ("cols", "<i4"), ("ld_nat", "<i4"), ("ld_t", "<i4"),
                           ("tile0", "<i4"), ("pad", "<i4")])
            assert dt.itemsize == 48
            segs = np.zeros(len(mats), dtype=dt)
            off, tile0, views = 0, 0, []
            for i, w in enumerate(mats):
                k, n = w.shape
                assert w.is_contiguous() and (w.data_ptr() - flat.data_ptr()) % 4 == 0
                rd, ld = (_padn(n), _pitch(k)) if transposed else (_padn(k), _pitch(n))
                segs[i] = ((w.data_ptr() - flat.data_ptr()) // 4, -1 if transposed else off, off if transposed else -1, k, n,
                           0 if transposed else ld, ld if transposed else 0, tile0, 0)
                views.append((off, rd * ld))
                off += (rd * ld + 127) // 128 * 128
                tile0 += ((k + 63) // 64) * ((n + 63) // 64)
            dst = self.t.zeros(off, dtype=self.t.bfloat16, device=self.dev)
            segs_d = self.t.from_numpy(segs.view(np.uint8).copy()).to(self.dev)
            plan = self._plans[key] = (segs_d, len(mats), tile0, dst, views)
        segs_d, nseg, tiles, dst, views = plan
        _lib.check(_L().lmrl_cast_bf16_segments(flat.data_ptr(), segs_d.data_ptr(), nseg, tiles, dst.data_ptr(), _sp()), "lmrl_cast_bf16_segments")
        tag = "wT" if transposed else "w"
        for w, (o, sz) in zip(mats, views):
            self.w[(tag, w.data_ptr())] = dst[o:o + sz]

    def _buf(self, name, numel):
        b = self.scratch.get(name)
        if b is None or b.numel() < numel:
            b = self.scratch[name] = self.t.empty(numel, dtype=self.t.bfloat16, device=self.dev)
        return b

    def cast(self, name, x, rows, cols, ld_src, transpose=False, keep=False, colsum=None):
        """K-major bf16 operand of `lmrl_gemm_bf16` from fp32 x [rows][cols] (row stride ld_src): [rows][pad64(cols)], or for
        transpose=True [pad64(cols)][pad64(rows)] = x^T; padding zero-filled.  keep=True: a per-step weight copy keyed by `name`.
        colsum=(out, accumulate) (transpose only): also out[c] (=|+=) sum_r x[r][c]."""
        if keep and name in self.w:
            return self.w[name]
        rd, ld = (_padn(rows), _pitch(cols)) if not transpose else (_padn(cols), _pitch(rows))
        dst = self.t.empty(rd * ld, dtype=self.t.bfloat16, device=self.dev) if keep else self._buf(name, rd * ld)
        if colsum is not None:
            assert transpose
            nws = _L().lmrl_cast_bf16_t_colsum_ws_bytes(rows, rd) // 2
            ws = self._buf("colsum_ws", nws)
            _lib.check(_L().lmrl_cast_bf16_t_colsum(x.data_ptr(), ld_src, rows, cols, dst.data_ptr(), ld, rd, colsum[0].data_ptr(), int(colsum[1]),
                                                    ws.data_ptr(), _sp()), "lmrl_cast_bf16_t_colsum")
            return dst
        _lib.check(_L().lmrl_cast_bf16(x.data_ptr(), ld_src, rows, cols, dst.data_ptr(), ld, rd, int(transpose), _sp()), "lmrl_cast_bf16")
        if keep:
            self.w[name] = dst
        return dst

    def transpose_staged(self, name, xb, ld_src, rows, cols, colsum=None):
        """[pad(cols)][pad(rows)] = xb^T from a bf16 operand xb [rows..][ld_src] a producer already staged; colsum as in `cast`."""
        rd, ld = _padn(cols), _pitch(rows)
        dst = self._buf(name, rd * ld)
        ws = None
        if colsum is not None:
            ws = self._buf("colsum_ws", _L().lmrl_cast_bf16_t_colsum_ws_bytes(rows, rd) // 2)
        _lib.check(_L().lmrl_transpose_bf16_colsum(xb.data_ptr(), ld_src, rows, cols, dst.data_ptr(), ld, rd,
                                                   colsum[0].data_ptr() if colsum is not None else None, int(colsum[1]) if colsum is not None else 0,
                                                   _lib.ptr(ws), _sp()), "lmrl_transpose_bf16_colsum")
        return dst

    def stage_dy(self, rows, n):
        """(buffer, row pitch) for a producer that writes the bf16 dy operand [pad(rows)][pitch(n)] of the next `linear_bwd` itself."""
        return self._buf("dy", _padn(rows) * _pitch(n)), _pitch(n)

    def stash(self, rows, k):
        """(fresh buffer, row pitch) for a staged operand [rows][k] that must outlive the next `stage`: the forward keeps its bf16 operands
        per layer, the backward's dW products read them as staged (`linear_bwd(xb=...)`: `lmrl_gemm_bf16_splitk_kmajor`, or a bf16 transpose where
        that kernel does not apply) — no fp32 copy of those activations exists."""
        assert k % 64 == 0
        return self.t.empty(_padn(rows) * _pitch(k), dtype=self.t.bfloat16, device=self.dev), _pitch(k)

    def stage(self, rows, k):
        """(buffer, row pitch) for a producer that writes the bf16 operand [rows][k] of the NEXT `linear_fwd` itself (k a multiple of 64)."""
        assert k % 64 == 0
        return self._buf("x", _padn(rows) * _pitch(k)), _pitch(k)

    def bias(self, b, n):
        """fp32 bias padded to a multiple of 64 entries (the GEMM epilogue reads whole 4-column groups)."""
        if n == _padn(n) and b.is_contiguous() and b.data_ptr() % 16 == 0:
            return b                        # already whole column groups: read in place (no per-step staging copy)
        key = ("bias", b.data_ptr())
        if key not in self.w:
            bp = self.t.zeros(_padn(n), dtype=self.t.float32, device=self.dev)
            bp[:n].copy_(b)
            self.w[key] = bp
        return self.w[key]

    def gemm(self, a, w, bias, c, m, n, k, ldc, n_store, accumulate=False, resid=None):
        """c[m][n_store] (=|+=) a[m][k] . w[n][k]^T + bias, fp32 out (lmrl_gemm_bf16 epilogues 3 / 2); a, w staged by `cast`
        (row pitch _pitch(k)).  resid (fp32 [m][ldc], another buffer): c = resid + a . w^T + bias in the same launch."""
        ld = _pitch(k)
        if resid is not None:
            assert not accumulate
            _lib.check(_L().lmrl_gemm_bf16_resid(a.data_ptr(), w.data_ptr(), _lib.ptr(bias), resid.data_ptr(), ldc, c.data_ptr(), m, n, _pad(k), ld, ld, ldc,
                                                 n_store, _sp()), "lmrl_gemm_bf16_resid")
            return
        if bias is None:       # few output tiles, long K (the dW products): split-K over ~2 workgroups per CU + a fixed-order reduce
            nws = _L().lmrl_gemm_bf16_splitk_ws_bytes(m, n, _pad(k))
            if nws:
                ws = self._buf("splitk_ws", nws // 2)
                _lib.check(_L().lmrl_gemm_bf16_splitk(a.data_ptr(), w.data_ptr(), c.data_ptr(), m, n, _pad(k), ld, ld, ldc, n_store, int(accumulate),
                                                      ws.data_ptr(), _sp()), "lmrl_gemm_bf16_splitk")
                return
        _lib.check(_L().lmrl_gemm_bf16_ld(a.data_ptr(), w.data_ptr(), _lib.ptr(bias), c.data_ptr(), m, n, _pad(k), ld, ld, ldc, n_store,
                                          2 if accumulate else 3, _sp()), "lmrl_gemm_bf16")


# bf16-matmul mode: Dense products whose epilogue writes the next kernel's bf16 operand (lmrl_gemm_bf16_gelu_dual / _qkv_heads / _gelu_bwd) instead of
# an elementwise pass over an [B*T][n] fp32 tensor.  A/B hook for tools/ and the second-path tests; the arithmetic differs from the unfused form
# only in gelu_new / its derivative being evaluated in the sigmoid form (v_exp + v_rcp, ~1e-7 relative) before the bf16 rounding.
FUSE_EPILOGUES = 7          # bit 0: c_attn -> staged q / k / v, bit 1: c_fc -> (pre-activation, bf16 gelu), bit 2: c_proj dX -> bf16 d(pre-activation)
FUSE_QKV, FUSE_GELU, FUSE_GELU_BWD = 1, 2, 4


def fused_ok(n: int, which: int) -> bool:
    return bool(int(FUSE_EPILOGUES) & which) and n % 128 == 0


# bf16-matmul mode: the c_fc pre-activation kept for the gelu backward is stored rounded to bf16 (its only reader is gelu'; the reference's
# `bf16_activations` keeps every activation in bf16): 100 instead of 201 MB per block written and read back at the ILQL batch.  False: fp32.
PRE_BF16 = True


def linear_fwd_gelu(mm: "MatmulBF16", xb, w, b, f, gb, ldg, rows, k, n):
    """f[rows][n] fp32 = x @ w + b and gb[rows][ldg] bf16 = gelu_new(f) in one launch (xb: the staged bf16 operand of x).  f None: only gb is
    written (a forward nobody differentiates).  f a bf16 tensor ([rows][pitch(n)]): the pre-activation rounded to bf16."""
    wt = mm.cast(("wT", w.data_ptr()), w, k, n, n, transpose=True, keep=True)
    if f is not None and f.dtype == mm.t.bfloat16:
        _lib.check(_L().lmrl_gemm_bf16_gelu_dual_prebf16(xb.data_ptr(), wt.data_ptr(), _lib.ptr(mm.bias(b, n)), f.data_ptr(), _pitch(n), gb.data_ptr(), ldg, rows, n,
                                                         _pad(k), _pitch(k), _pitch(k), _sp()), "lmrl_gemm_bf16_gelu_dual_prebf16")
        return
    _lib.check(_L().lmrl_gemm_bf16_gelu_dual(xb.data_ptr(), wt.data_ptr(), _lib.ptr(mm.bias(b, n)), _lib.ptr(f), n, gb.data_ptr(), ldg, rows, n, _pad(k),
                                             _pitch(k), _pitch(k), _sp()), "lmrl_gemm_bf16_gelu_dual")


def linear_fwd_qkv_heads(mm: "MatmulBF16", xb, w, b, flash_ws, rows, k, batch, heads, t):
    """attn.c_attn whose output goes straight into the flash kernels' staged q / k / v matrices in `flash_ws` (no fp32 qkv tensor)."""
    import ctypes
    n = 3 * heads * 64
    wt = mm.cast(("wT", w.data_ptr()), w, k, n, n, transpose=True, keep=True)
    q, plane = ctypes.c_void_p(), ctypes.c_long()
    _lib.check(_L().lmrl_flash_attn_stage_ptrs(flash_ws.data_ptr(), batch, heads, t, ctypes.byref(q), ctypes.byref(plane)), "lmrl_flash_attn_stage_ptrs")
    _lib.check(_L().lmrl_gemm_bf16_qkv_heads(xb.data_ptr(), wt.data_ptr(), _lib.ptr(mm.bias(b, n)), q, plane, rows, _pad(k), _pitch(k), _pitch(k), heads, t,
                                             _sp()), "lmrl_gemm_bf16_qkv_heads")
    _lib.check(_L().lmrl_flash_attn_finish_staging(flash_ws.data_ptr(), batch, heads, t, _sp()), "lmrl_flash_attn_finish_staging")


def linear_bwd_dx_gelu(mm: "MatmulBF16", dyb, w, pre, rows, k, n):
    """-> bf16 [pad(rows)][pitch(k)] = (dy @ w^T) * gelu_new'(pre): the dX product of mlp.c_proj fused with the gelu backward, written as the dy
    operand of the c_fc backward (`linear_bwd(dyb=...)`); dyb: the staged bf16 dy [rows][pitch(n)], w [k][n], pre fp32 [rows][k]."""
    wb = mm.cast(("w", w.data_ptr()), w, k, n, n, keep=True)                     # [k][pad(n)]
    dst, ldd = mm._buf("dy2", _padn(rows) * _pitch(k)), _pitch(k)
    if pre.dtype == mm.t.bfloat16:                                                # [rows][pitch(k)] bf16 (linear_fwd_gelu with a bf16 f)
        _lib.check(_L().lmrl_gemm_bf16_gelu_bwd_prebf16(dyb.data_ptr(), wb.data_ptr(), pre.data_ptr(), _pitch(k), dst.data_ptr(), ldd, rows, k, _pad(n), _pitch(n),
                                                        _pitch(n), _sp()), "lmrl_gemm_bf16_gelu_bwd_prebf16")
        return dst
    _lib.check(_L().lmrl_gemm_bf16_gelu_bwd(dyb.data_ptr(), wb.data_ptr(), pre.data_ptr(), k, dst.data_ptr(), ldd, rows, k, _pad(n), _pitch(n), _pitch(n),
                                            _sp()), "lmrl_gemm_bf16_gelu_bwd")
    return dst


# bf16-matmul mode, vocabulary-wide heads: logits written once in bf16 with the log-sum-exp partials and the target's fp32 logit out of the GEMM's
# registers (lmrl_gemm_bf16_ce), d(logits) formed in place (lmrl_ce_bwd_bf16_inplace) — no fp32 [rows][V] tensor, no lse pass over it.  The
# logits are rounded to bf16 ONCE; lse is the log-sum-exp of those stored values and the backward's softmax reads the same values, so its rows
# sum to 1 at any logit magnitude (the reference's bf16 mode: round once, then lse and softmax on the rounded logits); Q(s, a) / the target
# token's logit keep their fp32 accumulator value.
FUSE_CE = True


def head_fwd_ce(mm: "MatmulBF16", x, w, b, rows, k, n, targets, w_is_nk: bool = False, keep_logits: bool = True):
    """logits = x @ w + b for a vocabulary-wide head (w a Dense kernel [k][n]; w_is_nk: an embedding matrix [n][k], the tied LM head), never in
    fp32 -> (logits_bf16 [pad(rows)][pitch(n)] (fresh buffer), lse [rows], target logit [rows] fp32, target log-probability [rows]).
    keep_logits=False (inference: nobody differentiates this head): the logits are not stored at all (first result None)."""
    t = mm.t
    xb = mm.cast("x", x, rows, k, k)
    if w_is_nk:
        wt = mm.cast(("w", w.data_ptr()), w, n, k, k, keep=True)                     # [pad(n)][pitch(k)]
    else:
        wt = mm.cast(("wT", w.data_ptr()), w, k, n, n, transpose=True, keep=True)
    N = _padn(n)
    nslots = _L().lmrl_gemm_bf16_ce_slots(rows, N, _pad(k))
    yb = t.empty(_padn(rows) * _pitch(n), dtype=t.bfloat16, device=mm.dev) if keep_logits else None
    part = t.empty(rows * nslots * 2, dtype=t.float32, device=mm.dev)
    lse, tl, lp = (t.empty(rows, dtype=t.float32, device=mm.dev) for _ in range(3))
    _lib.check(_L().lmrl_gemm_bf16_ce(xb.data_ptr(), wt.data_ptr(), _lib.ptr(mm.bias(b, n) if b is not None else None), _lib.ptr(yb), _pitch(n), rows, N,
                                      _pad(k), _pitch(k), _pitch(k), n, targets.data_ptr(), tl.data_ptr(), part.data_ptr(), _sp()), "lmrl_gemm_bf16_ce")
    _lib.check(_L().lmrl_lse_from_partials(part.data_ptr(), nslots, rows, tl.data_ptr(), lse.data_ptr(), lp.data_ptr(), _sp()), "lmrl_lse_from_partials")
    return yb, lse, tl, lp


def ce_bwd_inplace(yb, n, lse, targets, coef_ce, coef_gather, rows):
    """the bf16 logits `yb` of `head_fwd_ce` -> the bf16 d(logits) operand of the head's backward products, in place"""
    _lib.check(_L().lmrl_ce_bwd_bf16_inplace(yb.data_ptr(), _pitch(n), n, lse.data_ptr(), targets.data_ptr(), _lib.ptr(coef_ce), _lib.ptr(coef_gather),
                                             rows, _padn(rows), _sp()), "lmrl_ce_bwd_bf16_inplace")
    return yb


# bf16-matmul mode: residual add inside the projection GEMM's epilogue (lmrl_gemm_bf16_resid) instead of a separate axpby launch.  Measured on
# one box, ILQL M3 step, A/B twice (tools/ab_train_resid.py, profiles/r03_train_resid_ab.txt): fused 46.93 / 47.31 ms, two launches 46.26 /
# 46.73 ms — the residual slice prefetched under the K loop costs the 128x128 / 256x256 tiles more registers than the 30 us axpby (150 MB at
# 5 TB/s) costs time.  Identical arithmetic (same loss to the last bit).  Kept as an A/B hook, OFF.
FUSE_RESIDUAL = False


def linear_fwd(x, w, b, y, rows, k, n, mm: Optional[MatmulBF16] = None, ldy=None, xb=None, resid=None, ldw=None):
    """y[rows][n] = x[rows][k] @ w[k][n] + b   (flax Dense / HF Conv1D kernel layout [in, out]); row stride of y = ldy (default n).
    xb: the bf16 operand of x if its producer already staged it (`MatmulBF16.stage`, LayerNorm / gelu / attention `_staged` forms).
    resid (fp32, same shape and pitch as y): y = resid + x @ w + b — the block's residual add inside the projection (one launch in the
    bf16-matmul mode; fp32 mode: the product, then one axpby, as before)."""
    ldy = ldy or n
    if mm is not None and resid is not None and not FUSE_RESIDUAL:      # A/B hook (tools/bench_train.py): the two-launch form
        linear_fwd(x, w, b, y, rows, k, n, mm=mm, ldy=ldy, xb=xb)
        axpby(1.0, y, 1.0, resid, y)
        return
    if mm is None:
        sgemm(x, w, y, rows, n, k, lda=k, ldb=ldw or n, ldc=ldy, bias=b)       # ldw: row pitch of w when it is a padded copy (fp32 mode)
        if resid is not None:
            axpby(1.0, y, 1.0, resid, y)
        return
    assert ldy % 4 == 0 and ldy >= _pad(n, 4), "bf16 matmul mode: the output row stride must cover whole 4-column groups"
    if xb is None:
        xb = mm.cast("x", x, rows, k, k)
    wt = mm.cast(("wT", w.data_ptr()), w, k, n, n, transpose=True, keep=True)          # [pad(n)][pad(k)]
    mm.gemm(xb, wt, mm.bias(b, n) if b is not None else None, y, rows, _padn(n), k, ldy, n, resid=resid)


def linear_bwd(x, w, dy, dx, dw, db, rows, k, n, ws, *, accumulate_dw=True, dx_beta=0.0, mm: Optional[MatmulBF16] = None, lddy=None, dyb=None,
               xb=None, ldw=None):
    """dx = dy @ w^T ; dw (+)= x^T @ dy ; db (+)= colsum(dy).  dyb (bf16 mode): dy already staged by its producer as the bf16 operand
    `mm.stage_dy(rows, n)` — then `dy` is not read (may be None) and the bias gradient sums the bf16 values.  xb (bf16 mode): the bf16
    operand of x the forward kept (`mm.stash(rows, k)`) — then `x` is not read (may be None)."""
    lddy = lddy or n
    if mm is None:
        if dx is not None:
            sgemm(dy, w, dx, rows, k, n, trans_b=True, lda=lddy, ldb=ldw or n, ldc=k, beta=dx_beta)
        if n == 1 and ws.numel() >= 64 * k:     # one output unit (value heads): x^T dy is a matrix-vector product, not 12 sgemm tiles over K = rows
            _lib.check(_L().lmrl_colsum_weighted(x.data_ptr(), rows, k, k, dy.data_ptr(), lddy, dw.data_ptr(), int(accumulate_dw), ws.data_ptr(), _sp()),
                       "lmrl_colsum_weighted")
        else:
            sgemm(x, dy, dw, k, n, rows, trans_a=True, lda=k, ldb=lddy, ldc=n, beta=1.0 if accumulate_dw else 0.0)
    else:
        assert dx_beta in (0.0, 1.0)
        if dx is not None:
            if dyb is None:
                dyb_ = mm.cast("dy", dy, rows, n, lddy)                                 # [rows][pad(n)]
            else:
                dyb_ = dyb
            assert k % 64 == 0, "bf16 matmul mode: layer widths must be multiples of 64"
            wb = mm.cast(("w", w.data_ptr()), w, k, n, n, keep=True)                     # [k][pad(n)]
            mm.gemm(dyb_, wb, None, dx, rows, k, n, k, k, accumulate=dx_beta == 1.0)
        if (FUSE_KMAJOR_DW and xb is not None and dyb is not None and k % 128 == 0 and n % 128 == 0 and rows % 64 == 0
                and _L().lmrl_gemm_bf16_splitk_ws_bytes(k, n, rows) > 0):
            # dW straight from the staged x [rows][pitch(k)] and dy [rows][pitch(n)] (no transposed copies); the bias gradient from dy's column sums
            nws = max(_L().lmrl_gemm_bf16_splitk_ws_bytes(k, n, rows), ((rows + 63) // 64) * n * 4)
            wsb = mm._buf("splitk_ws", nws // 2)
            if db is not None:
                _lib.check(_L().lmrl_colsum_bf16(dyb.data_ptr(), _pitch(n), rows, n, db.data_ptr(), int(accumulate_dw), wsb.data_ptr(), _sp()),
                           "lmrl_colsum_bf16")
            _lib.check(_L().lmrl_gemm_bf16_splitk_kmajor(xb.data_ptr(), dyb.data_ptr(), dw.data_ptr(), k, n, rows, _pitch(k), _pitch(n), n, n,
                                                         int(accumulate_dw), wsb.data_ptr(), _sp()), "lmrl_gemm_bf16_splitk_kmajor")
            return
        if xb is None:
            xt = mm.cast("xT", x, rows, k, k, transpose=True)                           # [pad(k)][pad(rows)]
        else:
            xt = mm.transpose_staged("xT", xb, _pitch(k), rows, k)
        cs = (db, accumulate_dw) if db is not None else None
        if dyb is None:
            dyt = mm.cast("dyT", dy, rows, n, lddy, transpose=True, colsum=cs)          # [pad(n)][pad(rows)]
        else:
            dyt = mm.transpose_staged("dyT", dyb, _pitch(n), rows, n, colsum=cs)
        db = None                                                                       # the bias gradient came out of the staging pass
        if n % 4 == 0:
            mm.gemm(xt, dyt, None, dw, k, _padn(n), rows, n, n, accumulate=accumulate_dw)
        else:   # rows of dw are not 16-byte aligned: produce dw^T [n][k] and add its transpose
            tmp = mm.t.empty(n, k, dtype=mm.t.float32, device=mm.dev)
            mm.gemm(dyt, xt, None, tmp, n, _pad(k), rows, k, k)
            _lib.check(_L().lmrl_transpose_add_f32(tmp.data_ptr(), k, dw.data_ptr(), n, n, k, 1.0 if accumulate_dw else 0.0, _sp()),
                       "lmrl_transpose_add_f32")
    if db is not None:
        colsum(dy, rows, n, lddy, db, accumulate_dw, ws)


def gather_dot(a, w, bias, idx, out, rows, k, n):
    """out[r] = a[r] . w[:, idx[r]] + bias[idx[r]]  (w [k][n] flax Dense kernel)"""
    _lib.check(_L().lmrl_gather_dot_f32(a.data_ptr(), k, w.data_ptr(), n, _lib.ptr(bias), idx.data_ptr(), out.data_ptr(), rows, k, n, _sp()),
               "lmrl_gather_dot_f32")


def colsum(x, rows, cols, ld, out, accumulate, ws):
    _lib.check(_L().lmrl_colsum(x.data_ptr(), rows, cols, ld, out.data_ptr(), int(accumulate), ws.data_ptr(), _sp()), "lmrl_colsum")


def gather_rows(src, idx, n, d):
    """dst[i] = src[idx[i]] (fp32 rows of d floats; idx int32 device tensor) -> new [n, d] tensor"""
    dst = src.new_empty((n, d))
    _lib.check(_L().lmrl_gather_rows_f32(src.data_ptr(), idx.data_ptr(), dst.data_ptr(), n, d, _sp()), "lmrl_gather_rows_f32")
    return dst


def scatter_rows(src, idx, dst, n, d, accumulate):
    """dst[idx[i]] (=|+=) src[i]; idx holds distinct rows"""
    _lib.check(_L().lmrl_scatter_rows_f32(src.data_ptr(), idx.data_ptr(), dst.data_ptr(), n, d, int(accumulate), _sp()), "lmrl_scatter_rows_f32")


def layernorm_fwd(x, g, b, y, mean, rstd, rows, d, eps):
    _lib.check(_L().lmrl_layernorm_fwd(x.data_ptr(), g.data_ptr(), b.data_ptr(), y.data_ptr(), mean.data_ptr(), rstd.data_ptr(), rows, d,
                                       float(eps), _sp()), "lmrl_layernorm_fwd")


def layernorm_fwd_staged(x, g, b, y, mean, rstd, yb, ldb, rows, d, eps):
    _lib.check(_L().lmrl_layernorm_fwd_staged(x.data_ptr(), g.data_ptr(), b.data_ptr(), _lib.ptr(y), mean.data_ptr(), rstd.data_ptr(), yb.data_ptr(), ldb,
                                              rows, d, float(eps), _sp()), "lmrl_layernorm_fwd_staged")


# the residual adds of a block folded into the LayerNorm behind them (lmrl_layernorm_add_fwd); False: one axpby launch per add (A/B hook)
FUSE_ADD_LN = True
# bf16-matmul mode: weight gradients from the operands as staged (lmrl_gemm_bf16_splitk_kmajor) instead of from transposed copies (A/B hook)
FUSE_KMAJOR_DW = True


def layernorm_add_fwd(x, resid, g, b, y, mean, rstd, yb, ldb, rows, d, eps):
    """x += resid (None: no add), then y / yb = LayerNorm(x) — one pass; x is updated in place."""
    _lib.check(_L().lmrl_layernorm_add_fwd(x.data_ptr(), _lib.ptr(resid), g.data_ptr(), b.data_ptr(), _lib.ptr(y), mean.data_ptr(), rstd.data_ptr(),
                                           _lib.ptr(yb), ldb, rows, d, float(eps), _sp()), "lmrl_layernorm_add_fwd")


def gelu_fwd_staged(x, y, yb, ldb, rows, cols):
    _lib.check(_L().lmrl_gelu_fwd_staged(x.data_ptr(), _lib.ptr(y), yb.data_ptr(), ldb, rows, cols, _sp()), "lmrl_gelu_fwd_staged")


def layernorm_bwd(dy, x, g, mean, rstd, dx, dy_xhat, rows, d, accumulate_dx):
    _lib.check(_L().lmrl_layernorm_bwd(dy.data_ptr(), x.data_ptr(), g.data_ptr(), mean.data_ptr(), rstd.data_ptr(), dx.data_ptr(),
                                       _lib.ptr(dy_xhat), rows, d, int(accumulate_dx), _sp()), "lmrl_layernorm_bwd")


def layernorm_bwd_fused_supported(d) -> bool:
    return bool(_L().lmrl_layernorm_bwd_fused_supported(int(d)))


def layernorm_bwd_fused_ws_floats(rows, d) -> int:
    return _L().lmrl_layernorm_bwd_fused_ws_bytes(int(rows), int(d)) // 4


def layernorm_bwd_fused(dy, x, g, mean, rstd, dx, dgamma, dbeta, rows, d, accumulate_dx, accumulate_dg, ws, dxb=None, ldb=0):
    """dx (+)= LN backward ; dgamma (+)= colsum(dy * xhat) ; dbeta (+)= colsum(dy) — one pass over the activations + a small reduce.
    dxb (bf16 mode): also the bf16 copy of the final dx (row pitch ldb), the dy operand of the next `linear_bwd(dyb=...)`."""
    _lib.check(_L().lmrl_layernorm_bwd_fused(dy.data_ptr(), x.data_ptr(), g.data_ptr(), mean.data_ptr(), rstd.data_ptr(), dx.data_ptr(),
                                             dgamma.data_ptr(), dbeta.data_ptr(), rows, d, int(accumulate_dx), int(accumulate_dg),
                                             ws.data_ptr(), _lib.ptr(dxb), ldb, _sp()), "lmrl_layernorm_bwd_fused")


def gelu_bwd_staged(mm, dy, x, rows, cols):
    """bf16-matmul mode: gelu backward written only as the bf16 dy operand of the next `linear_bwd(dyb=...)`"""
    dst, ldb = mm.stage_dy(rows, cols)
    _lib.check(_L().lmrl_gelu_bwd_bf16(dy.data_ptr(), x.data_ptr(), rows, cols, dst.data_ptr(), ldb, _padn(rows), _sp()), "lmrl_gelu_bwd_bf16")
    return dst


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


def adamw_segments(p, g, m, v, seg_end, seg_wd, lr, b1, b2, eps, step, target=None, alpha=0.0):
    """AdamW over a parameter arena in one launch; `target` (an arena of the same layout): the Polyak update target = alpha p_new + (1 - alpha)
    target in the same sweep (the coefficients are rounded to fp32 on the host exactly as the stand-alone `axpby` form passes them)."""
    _lib.check(_L().lmrl_adamw_segments_polyak(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), seg_end.data_ptr(), seg_wd.data_ptr(),
                                               seg_end.numel(), float(lr), float(b1), float(b2), float(eps), int(step),
                                               None if target is None else target.data_ptr(), float(alpha), float(1.0 - alpha), _sp()), "lmrl_adamw_segments")


def embed_fwd(wte, wpe, ids, pos, x, rows, d, vocab=0):
    """vocab > 0: rows of `wte`; ids outside [0, vocab) (a pad id beyond the model's vocabulary) embed as a zero row."""
    _lib.check(_L().lmrl_embed_fwd(wte.data_ptr(), wpe.data_ptr(), ids.data_ptr(), pos.data_ptr(), x.data_ptr(), rows, d, int(vocab), _sp()), "lmrl_embed_fwd")


def embed_bwd(dx, ids, pos, dwte, dwpe, rows, d, live=None, vocab=0, t_row=0):
    """live (uint8 [rows], optional): the attention mask — rows with flag 0 are skipped (their dx is exactly zero) unless, with t_row = T of the
    [B, T] batch, the next position of their sequence is attended (a loss term reads them then: left padding, masks with holes)."""
    _lib.check(_L().lmrl_embed_bwd(dx.data_ptr(), ids.data_ptr(), pos.data_ptr(), _lib.ptr(live), dwte.data_ptr(), dwpe.data_ptr(), rows, d, int(vocab),
                                   int(t_row), _sp()), "lmrl_embed_bwd")


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


def ce_bwd_staged(mm: MatmulBF16, logits, ld, vocab, lse, targets, coef_ce, coef_gather, rows):
    """bf16-matmul mode: d loss / d logits written straight into the bf16 dy operand of the head's backward products (no fp32 dlogits
    matrix, no cast pass over [rows][vocab]); returns the operand for `linear_bwd(dyb=...)` / `lm_head_backward(dlb=...)`."""
    dst, ldd = mm.stage_dy(rows, vocab)
    _lib.check(_L().lmrl_ce_bwd_bf16(logits.data_ptr(), ld, vocab, lse.data_ptr(), targets.data_ptr(), _lib.ptr(coef_ce), _lib.ptr(coef_gather),
                                     rows, dst.data_ptr(), ldd, _padn(rows), _sp()), "lmrl_ce_bwd_bf16")
    return dst


def mask_sum(sta, attn, n, out):
    _lib.check(_L().lmrl_mask_sum(_lib.ptr(sta), _lib.ptr(attn), n, out.data_ptr(), _sp()), "lmrl_mask_sum")


def flash_attn_ws(batch, heads, t, bf16, device):
    import torch
    L = _L()
    return (torch.empty(L.lmrl_flash_attn_ws_bytes(batch, heads, t, int(bf16)), dtype=torch.uint8, device=device),
            L.lmrl_flash_attn_lse_bytes(batch, heads, t) // 4)


def flash_attn_fwd(qkv, key_mask, att, lse, ws, batch, heads, t, bf16):
    _lib.check(_L().lmrl_flash_attn_fwd(qkv.data_ptr(), _lib.ptr(key_mask), att.data_ptr(), lse.data_ptr(), ws.data_ptr(), batch, heads, t, int(bf16),
                                        _sp()), "lmrl_flash_attn_fwd")


def flash_attn_fwd_staged(qkv, key_mask, att, lse, ws, att_b, ldb, batch, heads, t, bf16):
    """qkv None (bf16): `ws` already holds the staged q / k / v (`linear_fwd_qkv_heads`)."""
    _lib.check(_L().lmrl_flash_attn_fwd_staged(_lib.ptr(qkv), _lib.ptr(key_mask), _lib.ptr(att), lse.data_ptr(), ws.data_ptr(), att_b.data_ptr(), ldb,
                                               batch, heads, t, int(bf16), _sp()), "lmrl_flash_attn_fwd_staged")


def flash_attn_bwd(qkv, key_mask, att, datt, lse, dqkv, ws, batch, heads, t, bf16, qkv_staged=False):
    """qkv_staged: `ws` is the workspace the forward of these same qkv ran in and nothing has used it since — its staged q / k / v
    matrices are reused instead of staged again."""
    _lib.check(_L().lmrl_flash_attn_bwd(qkv.data_ptr(), _lib.ptr(key_mask), att.data_ptr(), datt.data_ptr(), lse.data_ptr(), dqkv.data_ptr(),
                                        ws.data_ptr(), batch, heads, t, int(bf16), int(qkv_staged), _sp()), "lmrl_flash_attn_bwd")


# bf16-matmul mode: D = rowsum(dO o O) of the flash backward from the bf16 attention output (the projection's operand) — no fp32 copy of O is written
# by the forward or read by the backward (50 MB per block each way at the ILQL batch).  False: the fp32 copy, as before.
D_FROM_BF16_O = True


def flash_attn_bwd_staged(mm, qkv, key_mask, att, datt, lse, ws, batch, heads, t, qkv_staged=False, attb=None, ld_attb=0):
    """bf16 kernels; d(qkv) written only as the bf16 dy operand of the c_attn `linear_bwd(dyb=...)`.  att None: D from the bf16 output `attb`."""
    dst, ldb = mm.stage_dy(batch * t, 3 * heads * 64)
    if att is None:
        _lib.check(_L().lmrl_flash_attn_bwd_staged_attb(_lib.ptr(qkv), _lib.ptr(key_mask), attb.data_ptr(), ld_attb, datt.data_ptr(), lse.data_ptr(), dst.data_ptr(),
                                                        ldb, ws.data_ptr(), batch, heads, t, int(qkv_staged), _sp()), "lmrl_flash_attn_bwd_staged_attb")
        return dst
    _lib.check(_L().lmrl_flash_attn_bwd_staged(_lib.ptr(qkv), _lib.ptr(key_mask), att.data_ptr(), datt.data_ptr(), lse.data_ptr(), dst.data_ptr(), ldb,
                                               ws.data_ptr(), batch, heads, t, int(qkv_staged), _sp()), "lmrl_flash_attn_bwd_staged")
    return dst
