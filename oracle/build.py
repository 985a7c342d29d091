"""Build recipe for the oracle's C helpers (test infrastructure only).

``python oracle/build.py`` -> oracle/_native.so.  Plain gcc, no fast-math, no FMA contraction, so
the sequential loops keep the reference's operation order.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "native.c")
OUT = os.path.join(HERE, "_native.so")


def build(force: bool = False) -> str:
    if not force and os.path.exists(OUT) and os.path.getmtime(OUT) >= os.path.getmtime(SRC):
        return OUT
    cmd = ["gcc", "-O2", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math", "-o", OUT, SRC, "-lm"]
    subprocess.run(cmd, check=True)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
