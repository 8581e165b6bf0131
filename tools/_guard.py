"""WMZ_GUARD_ALLOC=1: run a tool with every tensor at the end of its own hipMalloc region (tools/guard_alloc.cpp -> libguard_alloc.so)
so that a kernel running past the end of an operand faults.  Import before the first device allocation."""
import os, torch
if os.environ.get('WMZ_GUARD_ALLOC') in ('1', '2'):
    _lib = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'libguard_alloc.so')
    torch.cuda.memory.change_current_allocator(torch.cuda.memory.CUDAPluggableAllocator(_lib, 'guard_alloc', 'guard_free'))
    print('[guard allocator on]', flush=True)
