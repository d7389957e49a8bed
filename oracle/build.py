"""Build the C oracle (oracle/oracle.c -> oracle/_build/liboracle.so) with gcc.  Test infrastructure."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
OUT_DIR = os.path.join(HERE, "_build")
LIB = os.path.join(OUT_DIR, "liboracle.so")
SRC = os.path.join(HERE, "oracle.c")


def build(force=False):
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= os.path.getmtime(SRC):
        return LIB
    os.makedirs(OUT_DIR, exist_ok=True)
    # no -march=native: the .so built here travels to the GPU box whose host CPU may differ
    cmd = ["gcc", "-O3", "-fopenmp", "-shared", "-fPIC", "-o", LIB + ".tmp", SRC, "-lm"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0:
        raise RuntimeError("gcc failed:\n" + r.stdout.decode(errors="replace"))
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
