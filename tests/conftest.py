import contextlib
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    # WMZ_GUARD_ALLOC=1 (development sweep, GPU box): every tensor at the end of its own hipMalloc region (tools/guard_alloc.cpp), so a
    # kernel that runs past the end of an operand faults.  hipMalloc cannot run inside a stream capture: deselect the graph tests
    # (-k "not graph").  Must happen before the first device allocation of the process.
    if os.environ.get('WMZ_GUARD_ALLOC') in ('1', '2'):
        lib = os.path.join(ROOT, 'tools', 'libguard_alloc.so')
        alloc = torch.cuda.memory.CUDAPluggableAllocator(lib, 'guard_alloc', 'guard_free')
        torch.cuda.memory.change_current_allocator(alloc)

        def no_capture(self, *a, **k):          # a test that reaches a capture is skipped before the stream starts capturing
            pytest.skip('guard allocator: hipMalloc cannot run inside a stream capture')
        torch.cuda.CUDAGraph.capture_begin = no_capture


@pytest.fixture(autouse=True)
def _collect_after_each_test():
    """Captured hipGraphs (and the device memory of their private pools) held in reference cycles -- a model and the sampler
    session it carries, a trainer and its graph -- are let go when the test that made them is over, not at some later collection
    in the middle of another test's stream capture."""
    yield
    import gc
    gc.collect()


def load_golden(name):
    """tests/golden/<name>.npz -> dict of torch tensors (made by tests/golden/make_golden.py)."""
    with np.load(os.path.join(GOLDEN, name + '.npz')) as f:
        return {k: torch.from_numpy(f[k]) for k in f.files}


def sub(d, prefix):
    """Strip a 'prefix/' namespace from a golden dict."""
    n = len(prefix)
    return {k[n:]: v for k, v in d.items() if k.startswith(prefix)}


@pytest.fixture(scope='session')
def golden():
    return load_golden


def have_gpu():
    return torch.cuda.is_available()


def near_tie_mismatches(idx, idx_ref, latents_ref, codebook, rel_gap=1e-5):
    """VQ indices must be bit-identical to the reference's except at genuine near-ties.

    idx, idx_ref: integer tensors of the same shape; latents_ref: the ORACLE's (or the golden) latents [.., E] for those
    positions; codebook [C, E].  Every mismatching position must (i) have picked the oracle's second-best code and
    (ii) sit where the oracle's top-2 distance gap (d2 - d1) / d1 is below rel_gap (fp64).  Returns the mismatch count."""
    idx, idx_ref = idx.reshape(-1).cpu(), idx_ref.reshape(-1).cpu()
    bad = (idx != idx_ref).nonzero().reshape(-1)
    if bad.numel() == 0:
        return 0
    lat = latents_ref.reshape(-1, latents_ref.shape[-1]).double().cpu()[bad]
    d = (lat[:, None, :] - codebook.double().cpu()[None]).pow(2).sum(-1)
    top = d.topk(2, dim=-1, largest=False)
    gap = (top.values[:, 1] - top.values[:, 0]) / top.values[:, 0]
    print(f'[near-tie] {bad.numel()} of {idx.numel()} indices differ; oracle top-2 rel gaps: {gap.tolist()}')
    assert torch.equal(top.indices[:, 0], idx_ref[bad]), 'reference index is not the oracle argmin'
    assert torch.equal(top.indices[:, 1], idx[bad]), 'a differing index is not the runner-up code'
    assert float(gap.max()) < rel_gap, f'index differs away from a near-tie (gap {float(gap.max()):.3e})'
    return int(bad.numel())


@contextlib.contextmanager
def recorded_calls():
    """The C-ABI entry points a block of code reaches (their names, in order): tests assert WHICH kernels produced a result."""
    from world_modelz_amd import _lib
    seen = []
    orig = _lib.call

    def call(name, *a):
        seen.append(name)
        return orig(name, *a)
    _lib.call = call
    try:
        yield seen
    finally:
        _lib.call = orig


@contextlib.contextmanager
def chain_policy(policy):
    """config.chain_policy for the block ('always': the chain kernels whatever the token count -- parity tests run small grids,
    where 'auto' would take the op-by-op path; 'never': the op-by-op path)."""
    from world_modelz_amd import config
    prev = config.set_chain_policy(policy)
    try:
        yield
    finally:
        config.set_chain_policy(prev)
