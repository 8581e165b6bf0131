// Guard allocator for torch.cuda.memory.CUDAPluggableAllocator (development tool, not part of the product): every tensor gets
// its own hipMalloc region of whole 2 MB granules and sits at the END of it (256-byte aligned), so a kernel that reads or writes
// more than 255 bytes past the end of any tensor faults instead of silently touching a neighbour.
//   g++ -O2 -fPIC -shared -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include tools/guard_alloc.cpp -L/opt/rocm/lib -lamdhip64 -o tools/libguard_alloc.so
//   WMZ_GUARD_ALLOC=1 python -m pytest tests -m gpu -k "not graph" ...        (tests/conftest.py switches the allocator)
#include <hip/hip_runtime.h>
#include <sys/types.h>
#include <cstdlib>
#include <mutex>
#include <unordered_map>

static std::unordered_map<void*, void*> g_base;
static std::mutex g_mu;

extern "C" void* guard_alloc(ssize_t size, int device, hipStream_t) {
  if (size <= 0) size = 1;
  const size_t G = (size_t)2 << 20;
  const size_t tot = ((size_t)size + G - 1) / G * G;
  void* base = nullptr;
  (void)hipSetDevice(device);
  if (hipMalloc(&base, tot) != hipSuccess) return nullptr;
  // WMZ_GUARD_ALLOC=2: at the START of the region instead (what lies in front of a hipMalloc region is normally unmapped too):
  // catches reads / writes in FRONT of an operand
  static const bool at_start = [] { const char* e = getenv("WMZ_GUARD_ALLOC"); return e && e[0] == '2'; }();
  void* p = at_start ? base : (void*)((char*)base + ((tot - (size_t)size) & ~(size_t)255));
  std::lock_guard<std::mutex> l(g_mu);
  g_base[p] = base;
  return p;
}

extern "C" void guard_free(void* p, ssize_t, int, hipStream_t) {
  void* base = nullptr;
  {
    std::lock_guard<std::mutex> l(g_mu);
    auto it = g_base.find(p);
    if (it == g_base.end()) return;
    base = it->second;
    g_base.erase(it);
  }
  (void)hipFree(base);
}
