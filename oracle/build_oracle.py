"""Builds the oracle's compiled helpers (g++, host only) into oracle/_build/.  Test infrastructure: never imported by the product."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_build")
LIB = os.path.join(OUT, "libstdgamma.so")


def build(force: bool = False) -> str:
    src = os.path.join(HERE, "stdgamma.cpp")
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= os.path.getmtime(src):
        return LIB
    gxx = shutil.which("g++")
    if gxx is None:
        raise RuntimeError("g++ not found")
    os.makedirs(OUT, exist_ok=True)
    subprocess.run([gxx, "-O2", "-std=c++17", "-shared", "-fPIC", "-o", LIB, src], check=True)
    return LIB


if __name__ == "__main__":
    print(build(force=True))
