"""Build a timing variant of the library: one translation unit recompiled with extra -D flags, linked with the product's
other objects into tools/variants/libwmz_<tag>.so (load it with WMZ_LIB_PATH).

    python tools/build_variant.py <tag> <file.hip> -DWMZ_ATTN_ABL=7 [...]
"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from world_modelz_amd import build as B
tag, src, flags = sys.argv[1], sys.argv[2], sys.argv[3:]
B.build_library(verbose=False)
out = os.path.join(ROOT, 'tools', 'variants')
os.makedirs(out, exist_ok=True)
obj = os.path.join(out, f'{src[:-4]}_{tag}.o')
subprocess.run([B._hipcc()] + B.COMMON + B.PER_FILE.get(src, []) + flags + ['-c', os.path.join(B.CSRC, src), '-o', obj], check=True)
objs = [obj if s == src else os.path.join(B.OBJ, s[:-4] + '.o') for s in B.sources()]
lib = os.path.join(out, f'libwmz_{tag}.so')
subprocess.run([B._hipcc(), '--offload-arch=gfx950', '-shared', '-fPIC', '-o', lib] + objs, check=True)
print(lib)
