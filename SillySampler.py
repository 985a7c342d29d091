#!/usr/bin/env python3
"""UTAU / OpenUtau entry script: same command line and port-8572 protocol as GOOFER's SillySampler.py,
served by the MI355X backend (see goofer_amd/cli.py)."""
import sys

from goofer_amd.cli import main

if __name__ == "__main__":
    sys.exit(main())
