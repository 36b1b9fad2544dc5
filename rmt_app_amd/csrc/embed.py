#!/usr/bin/env python3
"""Wrap n2_kernels.inc into a C++ raw string literal (n2_kernels_embed.h) for rmt_n2.cpp."""
import sys
src, dst = sys.argv[1], sys.argv[2]
text = open(src).read()
assert ')RMTSRC"' not in text
with open(dst, "w") as f:
    f.write('R"RMTSRC(' + text + ')RMTSRC"\n')
