"""Build libwmz_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m world_modelz_amd.build [--force]

One object per .hip translation unit (compiled in parallel), linked into
world_modelz_amd/libwmz_hip.so.  Nothing is JIT-cached outside the tree, so the .so travels with
the repository snapshot to the GPU box.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
OBJ = os.path.join(HERE, 'build')
LIB = os.path.join(HERE, 'libwmz_hip.so')
ROOT = os.path.dirname(HERE)

COMMON = ['-O3', '--offload-arch=gfx950', '-fPIC', '-std=c++17', '-Wall', '-Wno-unused-function',
          '-Wno-unused-variable', '-Wno-unused-but-set-variable', '-Wno-unused-result', '-Wno-unused-lambda-capture',
          '-Wno-unused-local-typedef']
PER_FILE = {
    # distance arithmetic must round like the reference's separate sub/mul/add tensor ops
    'vq.hip': ['-ffp-contract=off'],
    'vq_screen.hip': ['-ffp-contract=off'],
}


def _hipcc():
    for c in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return 'hipcc'


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith('.hip'))


def _deps_mtime():
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')]
    hdrs.append(os.path.join(ROOT, 'include', 'wmz.h'))
    return max(os.path.getmtime(h) for h in hdrs)


def _hip_mtime(path, seen=None):
    """Newest mtime of a .hip source and of the .hip sources it #includes (transitively)."""
    seen = set() if seen is None else seen
    if path in seen or not os.path.exists(path):
        return 0.0
    seen.add(path)
    m = os.path.getmtime(path)
    with open(path) as f:
        for line in f:
            line = line.strip()
            if line.startswith('#include "') and line.endswith('.hip"'):
                m = max(m, _hip_mtime(os.path.join(CSRC, line[len('#include "'):-1]), seen))
    return m


def _compile(src, force, hdr_mtime):
    obj = os.path.join(OBJ, src[:-4] + '.o')
    path = os.path.join(CSRC, src)
    src_mtime = _hip_mtime(path)                # (a wrapper unit -- *_f16.hip, *_g<n>.hip -- is a second compilation of the source it includes)
    if (not force and os.path.exists(obj)
            and os.path.getmtime(obj) >= max(src_mtime, hdr_mtime)):
        return obj, False
    extra = os.environ.get('WMZ_EXTRA_HIPCC_FLAGS', '').split()     # kernel-tuning experiments (-DWMZ_FUSED_HPS=1 ...)
    cmd = [_hipcc()] + COMMON + PER_FILE.get(src, []) + extra + ['-c', path, '-o', obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f'hipcc failed for {src}:\n{r.stdout}\n{r.stderr}')
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    return obj, True


def build_library(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    hdr_mtime = _deps_mtime()
    srcs = sources()
    with ThreadPoolExecutor(max_workers=min(6, len(srcs))) as ex:
        res = list(ex.map(lambda s: _compile(s, force, hdr_mtime), srcs))
    objs = [o for o, _ in res]
    if force or any(c for _, c in res) or not os.path.exists(LIB):
        cmd = [_hipcc(), '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f'link failed:\n{r.stdout}\n{r.stderr}')
        if verbose:
            print(f'[wmz] built {LIB} from {len(objs)} objects')
    elif verbose:
        print(f'[wmz] {LIB} up to date')
    return LIB


if __name__ == '__main__':
    build_library(force='--force' in sys.argv)
