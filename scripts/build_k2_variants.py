"""Diagnostic builds of libmoma_hip.so with K2 compiled under extra -D flags (stamps / ablations); never the product.
    python scripts/build_k2_variants.py NAME=-DFLAG[,-DFLAG2] ...     ->  moma_amd/lib/variants/lib_NAME.so
Run one with MOMA_HIP_LIB=moma_amd/lib/variants/lib_NAME.so python scripts/ablate_k2.py"""
import os, subprocess, sys
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from moma_amd import build as B

B.build()
out = os.path.join(B.LIBDIR, "variants"); os.makedirs(out, exist_ok=True)
objdir = os.path.join(B.LIBDIR, "obj")
others = [os.path.join(objdir, f.replace(".hip", ".o")) for f in B._sources() if f != "infonce_fused.hip"]

def one(spec):
    name, flags = spec.split("=", 1)
    obj = os.path.join(out, f"infonce_{name}.o")
    subprocess.run([B.HIPCC, *B.CXXFLAGS, *[f for f in flags.split(",") if f], "-c", os.path.join(B.CSRC, "infonce_fused.hip"), "-o", obj], check=True)
    lib = os.path.join(out, f"lib_{name}.so")
    subprocess.run([B.HIPCC, "-shared", "-fPIC", f"--offload-arch={B.ARCH}", "-o", lib, obj, *others], check=True)
    os.remove(obj)
    print("built", lib, flush=True)

with ThreadPoolExecutor(4) as ex:
    list(ex.map(one, sys.argv[1:]))
