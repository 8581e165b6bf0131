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
    if len(_cache) > 2048:
        for k in [k for k, h in _cache.items() if any(r() is None for r in h[2])]:
            del _cache[k]
    _cache[key] = (ver, val, tuple(weakref.ref(p) for p in params))
    return val


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
    _cache[key] = (ver, val, tuple(weakref.ref(p) for p in params))
    return val


def invalidate():
    """Parameters changed in place without torch noticing (wmz_adamw_step on the flat arena)."""
    global _epoch
    _epoch += 1


def clear():
    _cache.clear()
