"""Operand-dtype copies of fp32 parameters, cached per (parameter version, dtype)."""
import weakref

import torch

_cache = {}
_epoch = 0      # bumped when parameters are rewritten behind torch's version counters (arena optimizer step)


def _key(ts, dtype, tag):
    return (tag, dtype) + tuple(id(t) for t in ts)


def operand(params, dtype, tag='w', build=None):
    """Return `build(*params)` (default: the single parameter itself) cast to `dtype`, contiguous, cached until
    any of the parameters is modified in place (optimizer step, load_state_dict) or re-allocated (.to())."""
    if not isinstance(params, (tuple, list)):
        params = (params,)
    if build is None and len(params) == 1 and params[0].dtype == dtype and params[0].is_contiguous():
        return params[0].detach()
    key = _key(params, dtype, tag)
    ver = (_epoch,) + tuple((p._version, p.data_ptr()) for p in params)
    hit = _cache.get(key)
    # the key is made of id()s: a freed parameter's id (and storage address, and version) can come back with another
    # model's parameter, so an entry only counts while its weak references still point at these very objects
    if hit is not None and hit[0] == ver and all(r() is p for r, p in zip(hit[2], params)):
        return hit[1]
    with torch.no_grad():
        src = build(*params) if build is not None else params[0]
        val = src.detach().to(dtype).contiguous()
    _sweep()
    _cache[key] = (ver, val, tuple(weakref.ref(p) for p in params))
    return val


_sweep_at = 256


def _sweep():
    """Drop the entries of parameters that no longer exist (a model that was deleted leaves ~10 MB of operand copies per 3 M
    parameters behind).  Runs when the cache has doubled since the last sweep (at least 256 entries): amortised O(1) per insert."""
    global _sweep_at
    if len(_cache) < _sweep_at:
        return
    for k in [k for k, h in _cache.items() if any(r() is None for r in h[2])]:
        del _cache[k]
    _sweep_at = max(256, 2 * len(_cache))


def cached(params, tag, build):
    """Like operand(), for values that are not a single cast tensor: `build(*params)` is kept until any parameter changes."""
    params = tuple(params)
    key = _key(params, None, tag)
    ver = (_epoch,) + tuple((p._version, p.data_ptr()) for p in params)
    hit = _cache.get(key)
    if hit is not None and hit[0] == ver and all(r() is p for r, p in zip(hit[2], params)):
        return hit[1]
    with torch.no_grad():
        val = build(*params)
    _sweep()
    _cache[key] = (ver, val, tuple(weakref.ref(p) for p in params))
    return val


_scoped = {}       # id(parameter) -> [weakref to it, how often invalidate(params) named it]
_unscoped = 0      # invalidate() calls that named nothing
_scoped_sweep_at = 1024


def invalidate(params=None):
    """Parameters changed in place without torch noticing (wmz_adamw_step on the flat arena).  Every operand copy is rebuilt at its
    next use either way (the global epoch moves); `params` says WHOSE weights moved, for holders of captured graphs: a graph
    re-captures only when its own model's tensors were named (graph.GraphedForward: the frozen auto-encoder's encoder graph must
    survive the denoiser's optimizer steps -- main.py:229-287 runs both in every training step)."""
    global _epoch, _unscoped, _scoped_sweep_at
    _epoch += 1
    if params is None:
        _unscoped += 1
        return
    for p in params:
        e = _scoped.get(id(p))
        # (the key is an id(): a freed parameter's id can come back with another model's tensor -- an entry counts only while its
        #  weak reference still points at this very object, like _cache's)
        if e is None or e[0]() is not p:
            _scoped[id(p)] = [weakref.ref(p), 1]
        else:
            e[1] += 1
    if len(_scoped) >= _scoped_sweep_at:
        for k in [k for k, e in _scoped.items() if e[0]() is None]:
            del _scoped[k]
        _scoped_sweep_at = max(1024, 2 * len(_scoped))


def epoch_of(tensors):
    """What a graph holder stamps: moves when invalidate() named one of `tensors` (or named nothing)."""
    n = 0
    for t in tensors:
        e = _scoped.get(id(t))
        if e is not None and e[0]() is t:
            n += e[1]
    return _unscoped, n


def clear():
    _cache.clear()


class BulkOperands:
    """Operand copies that are rebuilt together, by ONE launch of wmz_operands_refresh (training: after every optimizer
    step).  Each entry is registered under the same (params, dtype, tag) key the forward / backward code asks operand()
    for; refresh() fills the persistent destination tensors and stamps the cache entries valid for the current parameter
    versions, so those operand() calls hit.  Anything not registered, or invalidated in another way, still takes the
    ordinary per-operand path."""

    def __init__(self):
        self.entries = []          # (params, dtype, tag, transpose, zero_first, dst)

    def add(self, params, dtype, tag='w', transpose=False, zero_first=False):
        params = tuple(params)
        rows = sum(p.shape[0] for p in params) + (params[0].shape[0] if zero_first else 0)
        cols = params[0].numel() // params[0].shape[0]
        assert len(params) <= 2 and not (zero_first and len(params) != 1)
        shape = (cols, rows) if transpose else ((rows, cols) if params[0].dim() > 1 else (rows,))
        dst = torch.empty(shape, dtype=dtype, device=params[0].device)
        self.entries.append((params, dtype, tag, transpose, zero_first, dst))
        return dst

    def refresh(self):
        import ctypes
        from . import _lib as L
        for i0 in range(0, len(self.entries), 64):
            ent = self.entries[i0:i0 + 64]
            n = len(ent)
            vp, ci = ctypes.c_void_p * n, ctypes.c_int * n
            s0, s1, r0, r1, cc, dd, ff = vp(), vp(), ci(), ci(), ci(), vp(), ci()
            for i, (params, dtype, tag, tr, zf, dst) in enumerate(ent):
                a = params[0].detach()
                b = params[1].detach() if len(params) > 1 else None
                assert a.dtype == torch.float32 and a.is_contiguous() and (b is None or b.is_contiguous())
                if zf:
                    s0[i], s1[i], r0[i], r1[i] = None, a.data_ptr(), a.shape[0], a.shape[0]
                else:
                    s0[i], s1[i], r0[i], r1[i] = a.data_ptr(), (b.data_ptr() if b is not None else None), a.shape[0], (b.shape[0] if b is not None else 0)
                cc[i] = a.numel() // a.shape[0]
                dd[i] = dst.data_ptr()
                ff[i] = (1 if tr else 0) | (2 if dtype == torch.float32 else 0)
            L.call('wmz_operands_refresh', s0, s1, r0, r1, cc, dd, ff, n, L.stream())
        for params, dtype, tag, tr, zf, dst in self.entries:
            ver = (_epoch,) + tuple((p._version, p.data_ptr()) for p in params)
            _cache[_key(params, dtype, tag)] = (ver, dst, tuple(weakref.ref(p) for p in params))


class ConvOperands:
    """The conv encoder / decoder's GEMM operands (autoencoder.py: tags 'conv' and 'convT'), rebuilt together by ONE launch of
    wmz_conv_operands_refresh_packed after every optimizer step; refresh() stamps the cache entries valid for the current weights,
    so the forward / backward's operand() calls hit (built with tensor ops: permute, pad, flip, cast -- ~6 launches per layer).
    bf16: the same launch also writes the fragment-order weight streams of the direct kernels (csrc/conv_direct.hip,
    csrc/conv_point.hip: ops._direct_pack / ops._point_pack, cached per operand tensor) -- 34 pack launches per step otherwise."""

    def __init__(self, convs, dtype):
        self.dtype = dtype
        self.entries = []          # (weight, tag, mode, pack kind, dst, operand dst the pack belongs to | None)
        for conv in convs:
            w = conv.weight
            co, ci, kh, kw = w.shape
            ci8, co8 = (ci + 7) // 8 * 8, (co + 7) // 8 * 8
            for tag, mode, rows, kin in (('conv', 0, co8, ci8), ('convT', 1, ci8, co8)):
                op = torch.empty((rows, kh * kw * kin), dtype=dtype, device=w.device)
                self.entries.append((w, tag, mode, 0, op, None))
                if dtype != torch.bfloat16 or rows > 128 or rows % 8 != 0:
                    continue
                ncb = 2 if rows <= 64 else 4
                if kh == 3 and kw == 3 and kin in (64, 128):
                    self.entries.append((w, 'convq', mode, 1, torch.empty(9 * kin * ncb * 32, dtype=dtype, device=w.device), op))
                elif kh * kw * kin <= 256:
                    n = (kh * kw * kin + 63) // 64 * 64 * ncb * 32
                    self.entries.append((w, 'convp', mode, 2, torch.empty(n, dtype=dtype, device=w.device), op))

    def refresh(self):
        import ctypes
        from . import _lib as L
        for i0 in range(0, len(self.entries), 48):
            ent = self.entries[i0:i0 + 48]
            n = len(ent)
            vp, ci_ = ctypes.c_void_p * n, ctypes.c_int * n
            ws, ds, cos, cis, kks, ms, pk = vp(), vp(), ci_(), ci_(), ci_(), ci_(), ci_()
            for i, (w, tag, mode, pack, dst, _op) in enumerate(ent):
                wd = w.detach()
                assert wd.dtype == torch.float32 and wd.is_contiguous()
                ws[i], ds[i] = wd.data_ptr(), dst.data_ptr()
                cos[i], cis[i], kks[i], ms[i], pk[i] = w.shape[0], w.shape[1], w.shape[2] * w.shape[3], mode, pack
            L.call('wmz_conv_operands_refresh_packed', ws, ds, cos, cis, kks, ms, pk, n, L.dtype_code(self.dtype), L.stream())
        for w, tag, mode, pack, dst, op in self.entries:
            if pack == 0:
                ver = (_epoch, (w._version, w.data_ptr()))
                _cache[_key((w,), self.dtype, tag)] = (ver, dst, (weakref.ref(w),))
            else:               # keyed by the operand tensor it re-orders (ops._direct_pack / _point_pack -> cached((op,), tag, ..))
                ver = (_epoch, (op._version, op.data_ptr()))
                _cache[_key((op,), None, tag)] = (ver, dst, (weakref.ref(op),))
