#!/usr/bin/env python3
"""Builds librecnet_torch_ops.so (csrc/torch_ops.cpp: TORCH_LIBRARY(recnet, ...) over the C ABI) in-tree with g++.
Plain host C++ — no device code: it links against librecnet_hip.so (rpath $ORIGIN) and libtorch / libc10 / libc10_hip."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def build(force=False):
    import torch
    from torch.utils import cpp_extension as ce
    src = os.path.join(HERE, "torch_ops.cpp")
    out = os.path.join(HERE, "librecnet_torch_ops.so")
    deps = [src, os.path.join(HERE, "..", "..", "include", "recnet_hip.h"), os.path.join(HERE, "librecnet_hip.so")]
    if not force and os.path.exists(out) and all(os.path.getmtime(out) >= os.path.getmtime(d) for d in deps):
        return out
    tlib = os.path.join(os.path.dirname(torch.__file__), "lib")
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1",
           "-D_GLIBCXX_USE_CXX11_ABI=%d" % int(torch._C._GLIBCXX_USE_CXX11_ABI), "-Wno-deprecated-declarations"]
    for inc in ce.include_paths() + ["/opt/rocm/include"]:
        cmd += ["-isystem", inc]
    cmd += [src, "-o", out, "-L" + HERE, "-lrecnet_hip", "-L" + tlib, "-lc10", "-lc10_hip", "-ltorch_cpu", "-ltorch_hip", "-ltorch",
            "-Wl,-rpath,$ORIGIN", "-Wl,-rpath," + tlib]
    subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
