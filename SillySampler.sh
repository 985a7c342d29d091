#!/bin/sh
# UTAU / OpenUtau launcher of the MI355X resampler backend: the 13 resampler arguments go to SillySampler.py unchanged
# (no arguments: the port-8572 batch server).  PYTHON overrides the interpreter.
here=$(cd "$(dirname "$0")" && pwd -P)
exec "${PYTHON:-python3}" "$here/SillySampler.py" "$@"
